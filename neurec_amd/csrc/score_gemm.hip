// score_gemm.hip — evaluation scoring  S = P[users] · Qᵀ  on the fp32 matrix cores.
//
// Stands in for np.matmul(user_embed, item_embeddings.T)
// (model/general_recommender/MF.py:120-122, NGCF.py:149-150) and the TF
// matmul of LightGCN.predict (LightGCN.py:118-119,187-189).
//
// Numerics: every score is the k-ascending chain
//     acc = 0;  for k in 0..d-1: acc = fmaf(P[u][k], Q[i][k], acc)
// which is what v_mfma_f32_32x32x2_f32 computes bit for bit when the k-steps
// are issued in order (exact fp32 in, fp32 accumulate; no reduced precision).
// oracle/ restates the same chain on the CPU.
//
// Layout: both factors are first copied k-major (PT[k][user], QT[k][item]) so
// that an MFMA operand — lane l wants element [row l&31][k = 2s + (l>>5)] —
// is one coalesced 128-byte segment per half-wave, straight from L2 to a
// VGPR: no LDS staging, no barriers.  A wave keeps a 64-user A panel in
// registers and streams 64-item B tiles; the four waves of a block take
// four user panels against the same item tiles so the B lines are shared in
// L1.  The kernel is bound by the HBM write of S (4·I bytes per user), not by
// the matrix pipe: d is only 16..128.
#include "nr_common.h"
#include <algorithm>
#include <limits.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// dst[k][r] = (r < n && k < d) ? src[id(r)][k] : 0   for k < dp, r < npad
__global__ __launch_bounds__(256) void gather_transpose_kernel(
    const float* __restrict__ src, int64_t ld, const int32_t* __restrict__ ids, int n, int d,
    float* __restrict__ dst, int npad, int dp) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  // every load unconditional on a clamped (row, column), masked afterwards: a load behind a branch cannot be counted,
  // and the compiler then waits for each one before it issues the next (r06: this kernel was 16 x 2 dependent round
  // trips per thread, 13 us for a 10 MB table)
  const int k = k0 + tx, kc = min(k, d - 1);
  int64_t sr[16];
  float val[16];
  if (n > 0) {                                                  // (workgroup-uniform)
    if (ids) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) sr[rr] = (int64_t)ids[min(r0 + rr * 4 + ty, n - 1)];
    } else {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) sr[rr] = (int64_t)min(r0 + rr * 4 + ty, n - 1);
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) val[rr] = src[sr[rr] * ld + kc];
  }
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) tile[rr * 4 + ty][tx] = (n > 0 && r0 + rr * 4 + ty < n && k < d) ? val[rr] : 0.f;
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int k = k0 + kk * 4 + ty, r = r0 + tx;
    if (k < dp && r < npad) dst[(int64_t)k * npad + r] = tile[tx][kk * 4 + ty];
  }
}

// Operand-ordered copy of the k-major item table for the MFMA loops: QS[tile][x][s4][lane] (float4)
// holds, for lane = 32·h + j, the four B values QT[2·(4·s4 + e) + h][64·tile + 32·x + j], e = 0..3 —
// exactly what that lane feeds to four consecutive k-steps.  One coalesced 1 KB load per (x, s4).
__global__ __launch_bounds__(256) void swizzle_items_kernel(const float* __restrict__ QT, int ipad,
                                                            int ks, float4* __restrict__ QS,
                                                            int64_t n_vec) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= n_vec) return;
  const int lane = (int)(v & 63);
  const int64_t r = v >> 6;
  const int s4 = (int)(r % (ks / 4));
  const int64_t r2 = r / (ks / 4);
  const int x = (int)(r2 & 1);
  const int64_t t = r2 >> 1;
  const int j = lane & 31, h = lane >> 5;
  const float* q = QT + (int64_t)(8 * s4 + h) * ipad + t * 64 + x * 32 + j;
  QS[v] = make_float4(q[0], q[(int64_t)2 * ipad], q[(int64_t)4 * ipad], q[(int64_t)6 * ipad]);
}

template <int KS>
__global__ __launch_bounds__(256, 1) void score_gemm_kernel(
    const float* __restrict__ PT, int bpad, const float* __restrict__ QT, int ipad, int rows,
    float* __restrict__ S, int64_t lds, int wcols, int tiles_per_chunk) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int u0 = (blockIdx.x * 4 + wave) * 64;
  if (u0 >= bpad) return;

  float a0[KS], a1[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float* p = PT + (int64_t)(2 * s + h) * bpad + u0 + j;
    a0[s] = p[0];
    a1[s] = p[32];
  }
  const int n_tiles = ipad / 64;
  const int t_begin = blockIdx.y * tiles_per_chunk;
  const int t_end = min(n_tiles, t_begin + tiles_per_chunk);
  // B operands of the next tile are requested before the current tile's MFMAs are issued, so the
  // matrix pipe does not wait on L2 at every tile boundary (two register sets, tile loop unrolled
  // by two)
  float bA0[KS], bA1[KS], bB0[KS], bB1[KS];
  // B operands of a tile from the operand-ordered item copy (swizzle_items_kernel): 16-byte loads,
  // 2·KS/4 per tile instead of 2·KS.  Not a bandwidth matter: with 64 loads per tile and the next
  // tile's loads in flight, the first MFMA of a tile would have to wait with vmcnt(> 63) — the
  // counter has 6 bits, so it waited for the newest loads too and every tile stalled a memory
  // round trip (60 % of the MFMA peak; the ISA showed `s_waitcnt vmcnt(62)` before the first MFMA).
  auto load_b = [&](int t, float (&x0)[KS], float (&x1)[KS]) {
    const float4* q = reinterpret_cast<const float4*>(QT) + (int64_t)t * (2 * (KS / 4) * 64) + lane;
#pragma unroll
    for (int s4 = 0; s4 < KS / 4; ++s4) {
      const float4 v = q[s4 * 64], w = q[(KS / 4 + s4) * 64];
      x0[4 * s4] = v.x; x0[4 * s4 + 1] = v.y; x0[4 * s4 + 2] = v.z; x0[4 * s4 + 3] = v.w;
      x1[4 * s4] = w.x; x1[4 * s4 + 1] = w.y; x1[4 * s4 + 2] = w.z; x1[4 * s4 + 3] = w.w;
    }
  };
  auto tile = [&](int t, const float (&x0)[KS], const float (&x1)[KS]) {
    const int it = t * 64;
    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], x0[s], c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], x1[s], c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], x0[s], c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], x1[s], c11, 0, 0, 0);
    }
    // C/D map of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      const int ua = u0 + rr, ub = u0 + 32 + rr;
      const int ca = it + j, cb = it + 32 + j;
      if (ua < rows) {
        if (ca < wcols) S[(int64_t)ua * lds + ca] = c00[reg];
        if (cb < wcols) S[(int64_t)ua * lds + cb] = c01[reg];
      }
      if (ub < rows) {
        if (ca < wcols) S[(int64_t)ub * lds + ca] = c10[reg];
        if (cb < wcols) S[(int64_t)ub * lds + cb] = c11[reg];
      }
    }
  };
  if (t_begin >= t_end) return;
  // the loads of the tile after next are issued unconditionally (clamped to the chunk's last tile):
  // a load behind a branch cannot be counted, and `s_waitcnt vmcnt(0)` before the MFMAs would wait
  // for the tile just requested
  load_b(t_begin, bA0, bA1);
  for (int t = t_begin; t < t_end; t += 2) {
    load_b(min(t + 1, t_end - 1), bB0, bB1);
    tile(t, bA0, bA1);
    if (t + 1 < t_end) {
      load_b(min(t + 2, t_end - 1), bA0, bA1);
      tile(t + 1, bB0, bB1);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Tile maxima for the pruned evaluation: the same MFMA loop with the operands swapped (item tile
// as the row operand, user panel as the column operand), so that a lane holds 32 item scores of
// ONE user per 64-item tile and the tile maximum is an in-lane reduction.  The user's train items
// are struck out (-inf) before the maximum — a cursor per user column walks the ascending train
// list as the tiles go by — and so are the pad columns >= cols.  S is never written:
// M[user][tile] = max over the admissible items of 32-item tile `tile` (one MFMA row block),
// 4·2·⌈I/64⌉ bytes per user instead of 4·I.
// ---------------------------------------------------------------------------------------------
// maximum of the 16 accumulator values whose bit in `struck` is clear.  The struck positions are
// collected as bits (one or per strike) and applied inside the reduction: writing -inf into a
// run-time-indexed accumulator register costs a 16-way select per strike plus the AGPR round trip
// (the epilogue was ~800 instructions per tile against 128 MFMAs, a third of the kernel's time).
__device__ __forceinline__ float max16_except(const f32x16& c, uint32_t struck) {
  float m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) m = fmaxf(m, (struck >> r) & 1u ? -INFINITY : c[r]);
  return m;
}

// MASKED = false: the strikes are left to tilemax_fix_kernel (below) — no cursors, no per-tile strike code; only the
// pad columns of the last tile are excluded.  Every (user, tile) WITHOUT a train item already holds its final value.
template <int KS, bool MASKED>
__global__ __launch_bounds__(256, 1) void score_tilemax_kernel(
    const float* __restrict__ PT, int bpad, const float* __restrict__ QT, int ipad, int rows,
    int cols, const int32_t* __restrict__ users, const int64_t* __restrict__ tr_indptr,
    const int32_t* __restrict__ tr_indices, float* __restrict__ M, int64_t mld,
    int tiles_per_chunk) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int u0 = (blockIdx.x * 4 + wave) * 64;
  if (u0 >= bpad) return;

  float a0[KS], a1[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float* p = PT + (int64_t)(2 * s + h) * bpad + u0 + j;
    a0[s] = p[0];
    a1[s] = p[32];
  }
  const int n_tiles = ipad / 64;
  const int t_begin = blockIdx.y * tiles_per_chunk;
  const int t_end = min(n_tiles, t_begin + tiles_per_chunk);
  // train-list cursors of this lane's two users (column j of user block 0 / 1)
  const int ra = u0 + j, rb = u0 + 32 + j;
  int64_t pa = 0, ea = 0, pb = 0, eb = 0;
  if (MASKED && ra < rows) { const int64_t u = users ? users[ra] : ra; pa = tr_indptr[u]; ea = tr_indptr[u + 1]; }
  if (MASKED && rb < rows) { const int64_t u = users ? users[rb] : rb; pb = tr_indptr[u]; eb = tr_indptr[u + 1]; }
  if (MASKED) {
    // first position with item >= the chunk's first column, both users of the lane in lock step
    // (every probe is a memory round trip; the two searches one after the other were 10 % of a
    // block's time at 37 tiles per block)
    const int key = t_begin * 64;
    int64_t la = pa, ha = ea, lb = pb, hb = eb;
    while (__ballot(la < ha || lb < hb)) {
      const int64_t ma = (la + ha) >> 1, mb = (lb + hb) >> 1;
      const int va = la < ha ? tr_indices[ma] : 0, vb = lb < hb ? tr_indices[mb] : 0;
      if (la < ha) { if (va < key) la = ma + 1; else ha = ma; }
      if (lb < hb) { if (vb < key) lb = mb + 1; else hb = mb; }
    }
    pa = la;
    pb = lb;
  }
  int na = MASKED && pa < ea ? tr_indices[pa] : INT_MAX;      // next train item of each user
  int nb = MASKED && pb < eb ? tr_indices[pb] : INT_MAX;
  // ... and the one after it, requested a strike ahead: 94 % of the tiles hold a train item of one
  // of the wave's 64 users, and a cursor that loads its next item when it needs it puts a memory
  // round trip (1.5 us against 3.4 us of MFMAs) into every tile
  int na2 = MASKED && pa + 1 < ea ? tr_indices[pa + 1] : INT_MAX;
  int nb2 = MASKED && pb + 1 < eb ? tr_indices[pb + 1] : INT_MAX;

  float bA0[KS], bA1[KS], bB0[KS], bB1[KS];             // two B register sets (see score_gemm_kernel)
  // B operands of a tile from the operand-ordered item copy (swizzle_items_kernel): 16-byte loads,
  // 2·KS/4 per tile instead of 2·KS.  Not a bandwidth matter: with 64 loads per tile and the next
  // tile's loads in flight, the first MFMA of a tile would have to wait with vmcnt(> 63) — the
  // counter has 6 bits, so it waited for the newest loads too and every tile stalled a memory
  // round trip (60 % of the MFMA peak; the ISA showed `s_waitcnt vmcnt(62)` before the first MFMA).
  auto load_b = [&](int t, float (&x0)[KS], float (&x1)[KS]) {
    const float4* q = reinterpret_cast<const float4*>(QT) + (int64_t)t * (2 * (KS / 4) * 64) + lane;
#pragma unroll
    for (int s4 = 0; s4 < KS / 4; ++s4) {
      const float4 v = q[s4 * 64], w = q[(KS / 4 + s4) * 64];
      x0[4 * s4] = v.x; x0[4 * s4 + 1] = v.y; x0[4 * s4 + 2] = v.z; x0[4 * s4 + 3] = v.w;
      x1[4 * s4] = w.x; x1[4 * s4 + 1] = w.y; x1[4 * s4 + 2] = w.z; x1[4 * s4 + 3] = w.w;
    }
  };
  auto tile = [&](int t, const float (&x0)[KS], const float (&x1)[KS]) {
    const int it = t * 64;
    f32x16 c00 = {0}, c10 = {0}, c01 = {0}, c11 = {0};   // cXY: item block X (rows) x user block Y
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[s], a0[s], c00, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[s], a0[s], c10, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[s], a1[s], c01, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[s], a1[s], c11, 0, 0, 0);
    }
    // rows of the 32x32 result held by this lane: item = it + 32*X + (reg&3) + 8*(reg>>2) + 4*h
    // struck accumulator registers of user a / b: bits 0-15 item block 0, bits 16-31 item block 1
    uint32_t ka = 0u, kb = 0u;
    if (MASKED && __ballot(na < it + 64 || nb < it + 64)) {  // a train item of some user falls in the tile
      while (na < it + 64) {
        const int o = na - it, row = o & 31;
        if (((row >> 2) & 1) == h) ka |= 1u << ((row & 3) + 4 * (row >> 3) + (o < 32 ? 0 : 16));
        ++pa; na = na2; na2 = pa + 1 < ea ? tr_indices[pa + 1] : INT_MAX;
      }
      while (nb < it + 64) {
        const int o = nb - it, row = o & 31;
        if (((row >> 2) & 1) == h) kb |= 1u << ((row & 3) + 4 * (row >> 3) + (o < 32 ? 0 : 16));
        ++pb; nb = nb2; nb2 = pb + 1 < eb ? tr_indices[pb + 1] : INT_MAX;
      }
    }
    if (it + 64 > cols) {                                     // last tile: pad columns
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (it + rr >= cols) { ka |= 1u << reg; kb |= 1u << reg; }
        if (it + 32 + rr >= cols) { ka |= 1u << (16 + reg); kb |= 1u << (16 + reg); }
      }
    }
    // one maximum per 32-item half tile (= one MFMA row block): the two lane halves hold
    // different rows of the same block
    float ma0 = max16_except(c00, ka & 0xFFFFu), ma1 = max16_except(c10, ka >> 16);
    float mb0 = max16_except(c01, kb & 0xFFFFu), mb1 = max16_except(c11, kb >> 16);
    ma0 = fmaxf(ma0, __shfl_xor(ma0, 32, 64)); ma1 = fmaxf(ma1, __shfl_xor(ma1, 32, 64));
    mb0 = fmaxf(mb0, __shfl_xor(mb0, 32, 64)); mb1 = fmaxf(mb1, __shfl_xor(mb1, 32, 64));
    if (h == 0) {
      if (ra < rows) *reinterpret_cast<float2*>(M + (int64_t)ra * mld + 2 * t) = make_float2(ma0, ma1);
      if (rb < rows) *reinterpret_cast<float2*>(M + (int64_t)rb * mld + 2 * t) = make_float2(mb0, mb1);
    }
  };
  if (t_begin >= t_end) return;
  // the loads of the tile after next are issued unconditionally (clamped to the chunk's last tile):
  // a load behind a branch cannot be counted, and `s_waitcnt vmcnt(0)` before the MFMAs would wait
  // for the tile just requested
  load_b(t_begin, bA0, bA1);
  for (int t = t_begin; t < t_end; t += 2) {
    load_b(min(t + 1, t_end - 1), bB0, bB1);
    tile(t, bA0, bA1);
    if (t + 1 < t_end) {
      load_b(min(t + 2, t_end - 1), bA0, bA1);
      tile(t + 1, bB0, bB1);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Strikes as a separate, small pass.  The train matrix is fixed for an evaluator, so which (user, 32-item tile)
// pairs hold a train item — and which of the tile's 32 items they are — is known once: the PLAN, sorted by tile
// (neurec_amd/engine.py: TileStrikePlan; 1.03 M pairs at the gowalla shape against 38 M user-tiles).  The scoring
// loop then runs without cursors and strike code (score_tilemax_kernel<KS, false>), and this kernel recomputes just
// the planned pairs: one wave takes up to 32 planned users of ONE tile, gathers their factor rows, runs the same
// v_mfma_f32_32x32x2_f32 chain with the same operand roles (item tile = row operand, users = column operand; bit for
// bit the values of the scoring loop), strikes the planned items and the pad columns, and overwrites M[user][tile].
// ---------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void tilemax_fix_kernel(
    const float* __restrict__ P, int64_t ldp, int d, const float* __restrict__ QT, int ipad, int cols,
    const int32_t* __restrict__ chunk_tile, const int64_t* __restrict__ chunk_begin, int n_chunks,
    const int64_t* __restrict__ tile_ptr, const int32_t* __restrict__ plan_user,
    const uint32_t* __restrict__ plan_mask, const int32_t* __restrict__ row_of, int row_lo, int rows,
    float* __restrict__ M, int64_t mld) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int c = blockIdx.x * 4 + wave;
  if (c >= n_chunks) return;
  const int t = chunk_tile[c];
  const int64_t e = chunk_begin[c] + j, e_end = min(chunk_begin[c] + 32, tile_ptr[t + 1]);
  int u = 0, row = -1;
  uint32_t mask = 0u;
  if (e < e_end) {
    u = plan_user[e];
    mask = plan_mask[e];
    row = (row_of ? row_of[u] : u) - row_lo;
    if (row < 0 || row >= rows) row = -1;
  }
  if (!__ballot(row >= 0)) return;                             // none of these users is in this batch
  // operands: lane (j, h) feeds k = 2 s + h of item column j (A) and of user column j (B).  The 32 factor rows are
  // scattered: the wave fetches them one coalesced row per load (all 32 in flight) into its LDS slice and reads its
  // k-major operand values from there (row stride 2 KS + 1: conflict-free); a lane reading its own row two floats
  // apart made every load instruction touch 32 different lines.
  constexpr int DP = 2 * KS;
  __shared__ float sB[4][32][DP + 1];
  float a[KS], b[KS];
  const float* q = QT + (int64_t)h * ipad + (int64_t)t * 32 + j;
#pragma unroll
  for (int s = 0; s < KS; ++s) a[s] = q[(int64_t)2 * s * ipad];
  {
    float v0[32], v1[DP > 64 ? 32 : 1];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const float* pr = P + (int64_t)__shfl(u, r, 64) * ldp;
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
#pragma unroll
  for (int s = 0; s < KS; ++s) b[s] = sB[wave][j][2 * s + h];
  f32x16 acc = {0};
#pragma unroll
  for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
  // rows of the result held by this lane: item = 32 t + (reg & 3) + 8 (reg >> 2) + 4 h
  uint32_t struck = 0u;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int r = (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (((mask >> r) & 1u) || t * 32 + r >= cols) struck |= 1u << reg;
  }
  float m = max16_except(acc, struck);
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  if (h == 0 && row >= 0) M[(int64_t)row * mld + t] = m;
}

int padded_dim(int d) {
  const int opts[5] = {16, 32, 48, 64, 128};
  for (int i = 0; i < 5; ++i)
    if (d <= opts[i]) return opts[i];
  return -1;
}
inline int round_up64(int x) { return (x + 63) / 64 * 64; }

struct GemmWs {
  float* QT;     // k-major item copy (rescoring reads it)
  float* QS;     // operand-ordered item copy (the MFMA loops read it), same size
  float* PT;
  size_t qt_bytes, pt_bytes;   // qt_bytes covers both item copies
};
GemmWs carve(void* ws, int rows, int cols, int dp) {
  GemmWs g;
  const size_t one = nr_align_up((size_t)dp * round_up64(cols) * sizeof(float), 256);
  g.qt_bytes = 2 * one;
  g.pt_bytes = nr_align_up((size_t)dp * round_up64(rows > 0 ? rows : 1) * sizeof(float), 256);
  g.QT = (float*)ws;
  g.QS = (float*)((char*)ws + one);
  g.PT = (float*)((char*)ws + g.qt_bytes);
  return g;
}

// ---- the strike plan, built on the device (r05; was construction-time torch unique / bincount / cumsum) -------------
// A pair = (user, 32-item tile) holding at least one train item of the user.  CSR rows are sorted by item, so a
// user's entries of one tile are consecutive: the FIRST of them (the "head") stands for the pair.  Three launches:
// count heads per tile (LDS pre-aggregated histogram) -> one workgroup scans the tiles and lays out the chunk table
// (chunks of <= 32 pairs of one tile: the unit a wave of tilemax_fix_kernel recomputes) -> every head reserves a slot
// in its tile and writes (user, OR of its items' bits).  The order of the pairs inside a tile is whatever the atomics
// give — every pair is independent in the fix-up pass, so M does not depend on it.
// (a popular tile receives tens of thousands of heads: one global atomic each serialises — 0.31 ms at gowalla; a
// workgroup therefore counts its users' heads per tile in LDS first and publishes one atomic per (workgroup, tile))
constexpr int kPlanLdsTiles = 8192;             // tiles an LDS histogram holds (262,144 items); beyond: global atomics

template <bool LDS>
__global__ __launch_bounds__(256) void strike_plan_count_kernel(const int64_t* __restrict__ indptr,
                                                                const int32_t* __restrict__ indices, int n_users,
                                                                int n_tiles, int32_t* __restrict__ per_tile) {
  __shared__ int32_t s_cnt[LDS ? kPlanLdsTiles : 1];
  if (LDS) {
    for (int t = threadIdx.x; t < n_tiles; t += 256) s_cnt[t] = 0;
    __syncthreads();
  }
  // a workgroup owns a contiguous run of users (the fill kernel must see the same runs)
  const int per = (n_users + gridDim.x - 1) / gridDim.x;
  const int u0 = blockIdx.x * per, u1 = min(u0 + per, n_users);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int u = u0 + wave; u < u1; u += 4) {
    const int64_t b = indptr[u], e = indptr[u + 1];
    for (int64_t t = b + lane; t < e; t += NR_WAVE) {
      const int tile = indices[t] >> 5;
      if (t == b || (indices[t - 1] >> 5) != tile) atomicAdd(LDS ? &s_cnt[tile] : &per_tile[tile], 1);
    }
  }
  if (LDS) {
    __syncthreads();
    for (int t = threadIdx.x; t < n_tiles; t += 256)
      if (s_cnt[t]) atomicAdd(&per_tile[t], s_cnt[t]);
  }
}

__global__ __launch_bounds__(1024) void strike_plan_layout_kernel(const int32_t* __restrict__ per_tile, int n_tiles,
                                                                  int64_t* __restrict__ tile_ptr,
                                                                  int32_t* __restrict__ cursor,
                                                                  int32_t* __restrict__ chunk_tile,
                                                                  int64_t* __restrict__ chunk_begin,
                                                                  int32_t* __restrict__ counts) {
  __shared__ int64_t s_pairs[1024];
  __shared__ int32_t s_chunks[1024];
  const int tid = threadIdx.x;
  const int per = (n_tiles + 1023) / 1024, t0 = tid * per, t1 = min(t0 + per, n_tiles);
  int64_t pairs = 0;
  int32_t chunks = 0;
  for (int t = t0; t < t1; ++t) {
    pairs += per_tile[t];
    chunks += (per_tile[t] + 31) >> 5;
  }
  s_pairs[tid] = pairs;
  s_chunks[tid] = chunks;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {                       // inclusive scan of the per-thread totals
    const int64_t a = tid >= off ? s_pairs[tid - off] : 0;
    const int32_t c = tid >= off ? s_chunks[tid - off] : 0;
    __syncthreads();
    s_pairs[tid] += a;
    s_chunks[tid] += c;
    __syncthreads();
  }
  int64_t pbase = s_pairs[tid] - pairs;
  int32_t cbase = s_chunks[tid] - chunks;
  for (int t = t0; t < t1; ++t) {
    const int32_t n = per_tile[t];
    tile_ptr[t] = pbase;
    cursor[t] = 0;
    for (int c = 0; c < ((n + 31) >> 5); ++c) {
      chunk_tile[cbase + c] = t;
      chunk_begin[cbase + c] = pbase + 32 * (int64_t)c;
    }
    pbase += n;
    cbase += (n + 31) >> 5;
  }
  if (tid == 1023) {
    tile_ptr[n_tiles] = s_pairs[1023];
    counts[0] = (int32_t)s_pairs[1023];
    counts[1] = s_chunks[1023];
  }
}

template <bool LDS>
__global__ __launch_bounds__(256) void strike_plan_fill_kernel(const int64_t* __restrict__ indptr,
                                                               const int32_t* __restrict__ indices, int n_users,
                                                               int n_tiles, const int64_t* __restrict__ tile_ptr,
                                                               int32_t* __restrict__ cursor,
                                                               int32_t* __restrict__ plan_user,
                                                               uint32_t* __restrict__ plan_mask) {
  // LDS: pass 1 counts this workgroup's heads per tile, one global atomic per (workgroup, tile) reserves its slots
  // in the tile (s_base), pass 2 hands them out with LDS atomics
  __shared__ int32_t s_cnt[LDS ? kPlanLdsTiles : 1];
  __shared__ int32_t s_base[LDS ? kPlanLdsTiles : 1];
  const int per = (n_users + gridDim.x - 1) / gridDim.x;
  const int u0 = blockIdx.x * per, u1 = min(u0 + per, n_users);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (LDS) {
    for (int t = threadIdx.x; t < n_tiles; t += 256) s_cnt[t] = 0;
    __syncthreads();
    for (int u = u0 + wave; u < u1; u += 4) {
      const int64_t b = indptr[u], e = indptr[u + 1];
      for (int64_t t = b + lane; t < e; t += NR_WAVE) {
        const int tile = indices[t] >> 5;
        if (t == b || (indices[t - 1] >> 5) != tile) atomicAdd(&s_cnt[tile], 1);
      }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n_tiles; t += 256) {
      s_base[t] = s_cnt[t] ? atomicAdd(&cursor[t], s_cnt[t]) : 0;
      s_cnt[t] = 0;
    }
    __syncthreads();
  }
  for (int u = u0 + wave; u < u1; u += 4) {
    const int64_t b = indptr[u], e = indptr[u + 1];
    for (int64_t t = b + lane; t < e; t += NR_WAVE) {
      const int item = indices[t], tile = item >> 5;
      if (t != b && (indices[t - 1] >> 5) == tile) continue;     // not the head of its pair
      uint32_t m = 1u << (item & 31);
      for (int64_t q = t + 1; q < e && (indices[q] >> 5) == tile; ++q) m |= 1u << (indices[q] & 31);   // <= 31 more
      const int64_t slot = tile_ptr[tile] + (LDS ? s_base[tile] + atomicAdd(&s_cnt[tile], 1) : atomicAdd(&cursor[tile], 1));
      plan_user[slot] = u;
      plan_mask[slot] = m;
    }
  }
}

}  // namespace

extern "C" {

int nrhip_score_gemm_workspace_bytes(int rows, int cols, int d, size_t* bytes) {
  NR_REQUIRE(bytes && rows >= 0 && cols >= 1 && d >= 1, NR_ERR_ARG,
             "score_gemm_workspace_bytes: bad arguments");
  const int dp = padded_dim(d);
  NR_REQUIRE(dp > 0, NR_ERR_UNSUPPORTED, "score_gemm: embedding dim %d > 128 not built", d);
  GemmWs g = carve(nullptr, rows, cols, dp);
  *bytes = g.qt_bytes + g.pt_bytes;
  return NR_OK;
}

int nrhip_score_gemm_prepare_items(const float* d_Q, int64_t ldq, int cols, int d, void* d_ws,
                                   size_t ws_bytes, void* stream) {
  NR_REQUIRE(d_Q && d_ws && cols >= 1 && d >= 1 && ldq >= d, NR_ERR_ARG,
             "score_gemm_prepare_items: bad arguments");
  const int dp = padded_dim(d);
  NR_REQUIRE(dp > 0, NR_ERR_UNSUPPORTED, "score_gemm: embedding dim %d > 128 not built", d);
  GemmWs g = carve(d_ws, 0, cols, dp);
  NR_REQUIRE(ws_bytes >= g.qt_bytes, NR_ERR_WORKSPACE, "score_gemm_prepare_items: workspace small");
  const int ipad = round_up64(cols);
  hipLaunchKernelGGL(gather_transpose_kernel, dim3(ipad / 64, (dp + 63) / 64), dim3(256), 0,
                     (hipStream_t)stream, d_Q, ldq, (const int32_t*)nullptr, cols, d, g.QT, ipad,
                     dp);
  NR_LAUNCH_CHECK();
  const int64_t n_vec = (int64_t)(ipad / 64) * 2 * (dp / 8) * 64;      // ks = dp / 2 k-pairs per operand
  hipLaunchKernelGGL(swizzle_items_kernel, dim3((unsigned)((n_vec + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, g.QT, ipad, dp / 2, (float4*)g.QS, n_vec);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // extern "C"
// nrhip_score_gemm_prepare_items without the scoring loop's operand-ordered copy: the k-major copy is all the pruned
// evaluation reads when a bounded filter does the search (the fix-up and the rescoring; eval_pipeline.hip) — whoever
// scores with the fp32 loop afterwards (rows redone from full score rows) reloads the item side in full first
__attribute__((visibility("hidden"))) int nr_score_gemm_prepare_items_kmajor(const float* d_Q, int64_t ldq, int cols,
                                                                             int d, void* d_ws, size_t ws_bytes,
                                                                             void* stream) {
  NR_REQUIRE(d_Q && d_ws && cols >= 1 && d >= 1 && ldq >= d, NR_ERR_ARG, "score_gemm_prepare_items: bad arguments");
  const int dp = padded_dim(d);
  NR_REQUIRE(dp > 0, NR_ERR_UNSUPPORTED, "score_gemm: embedding dim %d > 128 not built", d);
  GemmWs g = carve(d_ws, 0, cols, dp);
  NR_REQUIRE(ws_bytes >= g.qt_bytes, NR_ERR_WORKSPACE, "score_gemm_prepare_items: workspace small");
  const int ipad = round_up64(cols);
  hipLaunchKernelGGL(gather_transpose_kernel, dim3(ipad / 64, (dp + 63) / 64), dim3(256), 0, (hipStream_t)stream, d_Q,
                     ldq, (const int32_t*)nullptr, cols, d, g.QT, ipad, dp);
  NR_LAUNCH_CHECK();
  return NR_OK;
}
// ... and the other half: the operand-ordered copy from a k-major copy that is already this table's (nrhip_eval_redo
// behind an evaluation that ran with prepare_items = 2)
__attribute__((visibility("hidden"))) int nr_score_gemm_swizzle_items(int cols, int d, void* d_ws, size_t ws_bytes,
                                                                      void* stream) {
  NR_REQUIRE(d_ws && cols >= 1 && d >= 1, NR_ERR_ARG, "score_gemm_swizzle_items: bad arguments");
  const int dp = padded_dim(d);
  NR_REQUIRE(dp > 0, NR_ERR_UNSUPPORTED, "score_gemm: embedding dim %d > 128 not built", d);
  GemmWs g = carve(d_ws, 0, cols, dp);
  NR_REQUIRE(ws_bytes >= g.qt_bytes, NR_ERR_WORKSPACE, "score_gemm_swizzle_items: workspace small");
  const int ipad = round_up64(cols);
  const int64_t n_vec = (int64_t)(ipad / 64) * 2 * (dp / 8) * 64;
  hipLaunchKernelGGL(swizzle_items_kernel, dim3((unsigned)((n_vec + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     g.QT, ipad, dp / 2, (float4*)g.QS, n_vec);
  NR_LAUNCH_CHECK();
  return NR_OK;
}
extern "C" {

int nrhip_score_gemm(const float* d_P, int64_t ldp, const int32_t* d_users, int rows, int cols,
                     int d, float* d_S, int64_t lds, void* d_ws, size_t ws_bytes, void* stream) {
  NR_REQUIRE(d_P && d_S && d_ws && cols >= 1 && d >= 1 && ldp >= d && lds >= cols && rows >= 0,
             NR_ERR_ARG, "score_gemm: bad arguments");
  const int dp = padded_dim(d);
  NR_REQUIRE(dp > 0, NR_ERR_UNSUPPORTED, "score_gemm: embedding dim %d > 128 not built", d);
  if (rows == 0) return NR_OK;
  GemmWs g = carve(d_ws, rows, cols, dp);
  NR_REQUIRE(ws_bytes >= g.qt_bytes + g.pt_bytes, NR_ERR_WORKSPACE,
             "score_gemm: workspace %zu < %zu", ws_bytes, g.qt_bytes + g.pt_bytes);
  hipStream_t st = (hipStream_t)stream;
  const int bpad = round_up64(rows), ipad = round_up64(cols);
  hipLaunchKernelGGL(gather_transpose_kernel, dim3(bpad / 64, (dp + 63) / 64), dim3(256), 0, st,
                     d_P, ldp, d_users, rows, d, g.PT, bpad, dp);
  NR_LAUNCH_CHECK();
  const int wcols = (int)(lds < (int64_t)ipad ? lds : (int64_t)ipad);
  const int bx = (bpad / 64 + 3) / 4;
  const int n_tiles = ipad / 64;
  // aim for >= 2048 blocks so all 256 CUs stay busy, but keep chunks >= 4 tiles
  int tpc = (int)(((int64_t)n_tiles * bx + 2047) / 2048);
  if (tpc < 4) tpc = 4;
  if (tpc > n_tiles) tpc = n_tiles;
  const int by = (n_tiles + tpc - 1) / tpc;
  dim3 grid(bx, by), block(256);
#define NR_GEMM_CASE(KS)                                                                    \
  hipLaunchKernelGGL(score_gemm_kernel<KS>, grid, block, 0, st, g.PT, bpad, g.QS, ipad, rows, \
                     d_S, lds, wcols, tpc)
  switch (dp) {
    case 16: NR_GEMM_CASE(8); break;
    case 32: NR_GEMM_CASE(16); break;
    case 48: NR_GEMM_CASE(24); break;
    case 64: NR_GEMM_CASE(32); break;
    case 128: NR_GEMM_CASE(64); break;
    default: NR_REQUIRE(false, NR_ERR_UNSUPPORTED, "score_gemm: dp=%d", dp);
  }
#undef NR_GEMM_CASE
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* The k-major item copy made by nrhip_score_gemm_prepare_items: QT[k][item], k < roundup(d),
 * item < *ipad (zero padded).  Used by the level-2 rescoring of the pruned evaluation. */
int nrhip_score_gemm_items_kmajor(const void* d_ws, int cols, int d, const float** qt, int* ipad) {
  NR_REQUIRE(d_ws && qt && ipad && cols >= 1, NR_ERR_ARG, "score_gemm_items_kmajor: bad arguments");
  NR_REQUIRE(padded_dim(d) > 0, NR_ERR_UNSUPPORTED, "score_gemm: embedding dim %d > 128 not built", d);
  *qt = (const float*)d_ws;
  *ipad = round_up64(cols);
  return NR_OK;
}

/* Pruned evaluation, level 1 (see nrhip_eval_tiles): M[r][t] = max over the admissible items of
 * 32-item tile t of the scores of user row r — train items of the user and columns >= cols
 * excluded — computed by the scoring loop without ever writing the scores.  Item side prepared
 * with nrhip_score_gemm_prepare_items; d_M has rows x mld floats, mld even, >= 2*ceil(cols/64)
 * (tiles beyond ceil(cols/32) hold -inf). */
int nrhip_score_tilemax(const float* d_P, int64_t ldp, const int32_t* d_users, int rows, int cols,
                        int d, const int64_t* d_tr_indptr, const int32_t* d_tr_indices,
                        float* d_M, int64_t mld, void* d_ws, size_t ws_bytes, void* stream) {
  NR_REQUIRE(d_P && d_M && d_ws && (!d_tr_indptr == !d_tr_indices) && cols >= 1 && d >= 1 && ldp >= d &&
                 rows >= 0 && mld >= 2 * ((cols + 63) / 64) && mld % 2 == 0,
             NR_ERR_ARG, "score_tilemax: bad arguments (mld must be even and >= 2*ceil(cols/64))");
  const bool masked = d_tr_indptr != nullptr;
  const int dp = padded_dim(d);
  NR_REQUIRE(dp > 0, NR_ERR_UNSUPPORTED, "score_tilemax: embedding dim %d > 128 not built", d);
  if (rows == 0) return NR_OK;
  GemmWs g = carve(d_ws, rows, cols, dp);
  NR_REQUIRE(ws_bytes >= g.qt_bytes + g.pt_bytes, NR_ERR_WORKSPACE,
             "score_tilemax: workspace %zu < %zu", ws_bytes, g.qt_bytes + g.pt_bytes);
  hipStream_t st = (hipStream_t)stream;
  const int bpad = round_up64(rows), ipad = round_up64(cols);
  hipLaunchKernelGGL(gather_transpose_kernel, dim3(bpad / 64, (dp + 63) / 64), dim3(256), 0, st,
                     d_P, ldp, d_users, rows, d, g.PT, bpad, dp);
  NR_LAUNCH_CHECK();
  const int bx = (bpad / 64 + 3) / 4;
  const int n_tiles = ipad / 64;
  int tpc = (int)(((int64_t)n_tiles * bx + 2047) / 2048);
  if (tpc < 4) tpc = 4;
  if (tpc > n_tiles) tpc = n_tiles;
  const int by = (n_tiles + tpc - 1) / tpc;
  dim3 grid(bx, by), block(256);
#define NR_TMAX_CASE(KS)                                                                                     \
  if (masked)                                                                                                \
    hipLaunchKernelGGL((score_tilemax_kernel<KS, true>), grid, block, 0, st, g.PT, bpad, g.QS, ipad, rows,  \
                       cols, d_users, d_tr_indptr, d_tr_indices, d_M, mld, tpc);                            \
  else                                                                                                       \
    hipLaunchKernelGGL((score_tilemax_kernel<KS, false>), grid, block, 0, st, g.PT, bpad, g.QS, ipad, rows, \
                       cols, d_users, d_tr_indptr, d_tr_indices, d_M, mld, tpc)
  switch (dp) {
    case 16: NR_TMAX_CASE(8); break;
    case 32: NR_TMAX_CASE(16); break;
    case 48: NR_TMAX_CASE(24); break;
    case 64: NR_TMAX_CASE(32); break;
    case 128: NR_TMAX_CASE(64); break;
    default: NR_REQUIRE(false, NR_ERR_UNSUPPORTED, "score_tilemax: dp=%d", dp);
  }
#undef NR_TMAX_CASE
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* Level 1, second half, when nrhip_score_tilemax ran WITHOUT the train lists (d_tr_indptr = d_tr_indices = NULL):
 * recompute M[row][tile] for the planned (user, tile) pairs with the planned items struck.  The plan (built once
 * per train matrix, neurec_amd/engine.py: TileStrikePlan): pairs sorted by 32-item tile — d_tile_ptr[n_tiles32 + 1],
 * d_plan_user[e], d_plan_mask[e] (bit r: item 32*tile + r is a train item of the user) — cut into chunks of <= 32
 * pairs of one tile (d_chunk_tile[c], d_chunk_begin[c]).  d_row_of[user] = row of the user in the evaluation order
 * (or -1; NULL: row = user); rows [row_lo, row_lo + rows) are the batch M holds.  After this call M equals what
 * nrhip_score_tilemax computes WITH the train lists, bit for bit. */
int nrhip_score_tilemax_fix(const float* d_P, int64_t ldp, int d, int cols, const int32_t* d_chunk_tile,
                            const int64_t* d_chunk_begin, int n_chunks, const int64_t* d_tile_ptr,
                            const int32_t* d_plan_user, const uint32_t* d_plan_mask, const int32_t* d_row_of,
                            int row_lo, int rows, float* d_M, int64_t mld, const void* d_ws, size_t ws_bytes,
                            void* stream) {
  NR_REQUIRE(d_P && d_M && d_ws && d_tile_ptr && cols >= 1 && d >= 1 && ldp >= d && rows >= 0 && n_chunks >= 0 &&
                 mld >= 2 * ((cols + 63) / 64), NR_ERR_ARG, "score_tilemax_fix: bad arguments");
  if (rows == 0 || n_chunks == 0) return NR_OK;
  NR_REQUIRE(d_chunk_tile && d_chunk_begin && d_plan_user && d_plan_mask, NR_ERR_ARG, "score_tilemax_fix: plan");
  const int dp = padded_dim(d);
  NR_REQUIRE(dp > 0, NR_ERR_UNSUPPORTED, "score_tilemax_fix: embedding dim %d > 128 not built", d);
  GemmWs g = carve(const_cast<void*>(d_ws), 0, cols, dp);
  NR_REQUIRE(ws_bytes >= g.qt_bytes, NR_ERR_WORKSPACE, "score_tilemax_fix: workspace %zu < %zu", ws_bytes, g.qt_bytes);
  const int ipad = round_up64(cols);
  dim3 grid((n_chunks + 3) / 4), block(256);
#define NR_FIX_CASE(KS)                                                                                          \
  hipLaunchKernelGGL(tilemax_fix_kernel<KS>, grid, block, 0, (hipStream_t)stream, d_P, ldp, d, g.QT, ipad, cols, \
                     d_chunk_tile, d_chunk_begin, n_chunks, d_tile_ptr, d_plan_user, d_plan_mask, d_row_of,    \
                     row_lo, rows, d_M, mld)
  switch (dp) {
    case 16: NR_FIX_CASE(8); break;
    case 32: NR_FIX_CASE(16); break;
    case 48: NR_FIX_CASE(24); break;
    case 64: NR_FIX_CASE(32); break;
    case 128: NR_FIX_CASE(64); break;
    default: NR_REQUIRE(false, NR_ERR_UNSUPPORTED, "score_tilemax_fix: dp=%d", dp);
  }
#undef NR_FIX_CASE
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* The strike plan of a train matrix (engine.TileStrikePlan; uni_evaluator.py:132-140 strikes the train items of every
 * batch on the host): which (user, 32-item tile) pairs hold a train item and which items of the tile, grouped by
 * tile and cut into chunks of <= 32 pairs — the arguments of nrhip_score_tilemax_fix.  Capacities: d_plan_user /
 * d_plan_mask nnz entries, d_tile_ptr n_tiles + 1 (n_tiles = 2 * ceil(cols / 64)), d_chunk_tile / d_chunk_begin
 * nnz / 32 + n_tiles + 1, d_counts[2] = {pairs, chunks}, d_ws 2 * n_tiles int32.  Built once per train matrix. */
int nrhip_tile_strike_plan(const int64_t* d_indptr, const int32_t* d_indices, int n_users, int cols,
                           int32_t* d_plan_user, uint32_t* d_plan_mask, int64_t* d_tile_ptr, int32_t* d_chunk_tile,
                           int64_t* d_chunk_begin, int32_t* d_counts, void* d_ws, size_t ws_bytes, void* stream) {
  NR_REQUIRE(d_indptr && d_indices && d_plan_user && d_plan_mask && d_tile_ptr && d_chunk_tile && d_chunk_begin &&
                 d_counts && d_ws && n_users >= 0 && cols >= 1, NR_ERR_ARG, "tile_strike_plan: bad arguments");
  const int n_tiles = 2 * ((cols + 63) / 64);
  NR_REQUIRE(ws_bytes >= (size_t)n_tiles * 8, NR_ERR_WORKSPACE, "tile_strike_plan: workspace %zu < %zu", ws_bytes,
             (size_t)n_tiles * 8);
  hipStream_t st = (hipStream_t)stream;
  int32_t* per_tile = (int32_t*)d_ws;
  int32_t* cursor = per_tile + n_tiles;
  NR_CHECK_HIP(hipMemsetAsync(per_tile, 0, (size_t)n_tiles * 4, st));
  // both passes cut the users into the same contiguous runs, one per workgroup (two resident per CU)
  const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(((int64_t)n_users + 15) / 16, 512));
  const bool lds = n_tiles <= kPlanLdsTiles;
  if (lds)
    hipLaunchKernelGGL(strike_plan_count_kernel<true>, dim3(blocks), dim3(256), 0, st, d_indptr, d_indices, n_users,
                       n_tiles, per_tile);
  else
    hipLaunchKernelGGL(strike_plan_count_kernel<false>, dim3(blocks), dim3(256), 0, st, d_indptr, d_indices, n_users,
                       n_tiles, per_tile);
  NR_LAUNCH_CHECK();
  hipLaunchKernelGGL(strike_plan_layout_kernel, dim3(1), dim3(1024), 0, st, per_tile, n_tiles, d_tile_ptr, cursor,
                     d_chunk_tile, d_chunk_begin, d_counts);
  NR_LAUNCH_CHECK();
  if (lds)
    hipLaunchKernelGGL(strike_plan_fill_kernel<true>, dim3(blocks), dim3(256), 0, st, d_indptr, d_indices, n_users,
                       n_tiles, d_tile_ptr, cursor, d_plan_user, d_plan_mask);
  else
    hipLaunchKernelGGL(strike_plan_fill_kernel<false>, dim3(blocks), dim3(256), 0, st, d_indptr, d_indices, n_users,
                       n_tiles, d_tile_ptr, cursor, d_plan_user, d_plan_mask);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // extern "C"
