// gemm.hip — a general fp32 GEMM on the matrix cores for the dense products that are NOT the evaluation's
// scoring loop: the wide Mult-VAE shapes (conf/MultiVAE.properties:3 lists p_dim [200, 600]; MultiVAE.py:73-135:
// tf.matmul in q_graph / p_graph and their autodiff) whose hidden widths do not fit the register-resident 16/32-wide
// forms of vae.hip.
//
//     C[m][n] (+)= sum_k A[k][m] * B[k][n]          A: [K][lda], B: [K][ldb]  ("k-major": the contraction index is the
//                                                    slow one), C: [M][ldc] row-major
// Either operand may instead be given "k-minor" — A as [M][lda], B as [N][ldb], the contraction index contiguous (a
// row-major activation block as the left factor; a TF weight [in][out] as the right factor of a product with its
// transpose): the tile is then fetched as 64-byte row pieces and laid k-major into LDS on the way, so no caller
// transposes anything.
//
// k-major on both sides is the layout in which an operand of v_mfma_f32_32x32x2_f32 — lane l feeds element
// [row or column l & 31][k = l >> 5] — is one coalesced 128-byte segment per half-wave straight from L2 (the layout
// score_gemm.hip transposes its factors into).  The callers pick which of their arrays already are k-major:
//     logits  S[b][i]   = sum_k g^T[k][b] * W[k][i]      A = g^T (feature-major activations), B = the TF variable [h][I]
//     dW[k][i]          = sum_b g[b][k]   * D[b][i]      A = g row-major, B = dLoss/dlogits row-major — no transposes
//     dg[b][k]          = sum_i D^T[i][b] * W^T[i][k]    the one product that needs both operands transposed first
// Numerics: every C element is the k-ascending chain acc = fmaf(A[k][m], B[k][n], acc) — what the MFMA evaluates when
// its k-steps are issued in order (score_gemm.hip; checked bit for bit there) — continued from C when `accumulate`.
// A split over K (long contractions with a small M x N: dg above) writes one partial product per split and adds them
// in split order: deterministic, re-associated at the split boundaries only.
#include "nr_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KT = 16;           // k rows per LDS tile

__device__ __forceinline__ float epilogue_act(int a, float x) {
  if (a == 0) return tanhf(x);
  if (a == 1) return 1.0f / (1.0f + expf(-x));
  if (a == 2) return fmaxf(x, 0.f);
  return x;
}

// Block = 4 waves (2 x 2) over a TILE x TILE tile of C; wave tile (TILE/2)^2 = R x R MFMA 32x32 tiles.
// Both operand tiles [KT][TILE] go global -> registers -> LDS (double-buffered: the next tile's loads are in flight
// while the matrix cores work on the current one); an MFMA operand is then one conflict-free 32-bank row read.
// The linear block id is dealt to the 8 XCDs so that the tiles_m blocks sharing one n-tile of B run on the SAME XCD
// (one L2 fetch of that B tile serves all of them).
template <int TILE, bool ACC, bool AKM, bool BKM>      // AKM / BKM: operand A / B is k-major (else k-minor)
__global__ __launch_bounds__(256, 3) void gemm_lds_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, int M, int N, int K,
    int k_per_split, float* __restrict__ C, int64_t ldc, int64_t split_stride, const float* __restrict__ bias_n,
    int act, int tiles_m, int tiles_n) {
  constexpr int R = TILE / 64;
  constexpr int PER = KT * TILE / 256;
  // rows padded by one float: the k-minor stores (16 lanes = 16 k of one column) then fall on 16 different banks
  __shared__ float sA[2][KT][TILE + 1];
  __shared__ float sB[2][KT][TILE + 1];
  const int total = tiles_m * tiles_n, per_xcd = (total + 7) / 8;
  // (rotated by the split index: with fewer than 8 tiles the same x — the same XCD — would do every split's work)
  const unsigned bx = (blockIdx.x + blockIdx.z) % gridDim.x;
  const int logical = (bx % 8) * per_xcd + bx / 8;
  if (logical >= total) return;
  const int m0 = (logical % tiles_m) * TILE, n0 = (logical / tiles_m) * TILE;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;
  const int wm = (wave & 1) * (TILE / 2), wn = (wave >> 1) * (TILE / 2);
  const int kb = blockIdx.z * k_per_split, ke = min(K, kb + k_per_split);
  C += (int64_t)blockIdx.z * split_stride;

  f32x16 acc[R][R];
#pragma unroll
  for (int x = 0; x < R; ++x)
#pragma unroll
    for (int y = 0; y < R; ++y) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        float v = 0.f;
        if (ACC) {
          const int r = m0 + wm + 32 * x + (reg & 3) + 8 * (reg >> 2) + 4 * h, c = n0 + wn + 32 * y + j;
          if (r < M && c < N) v = C[(int64_t)r * ldc + c];
        }
        acc[x][y][reg] = v;
      }
    }
  if (kb >= ke) return;                                   // an empty split (never launched; kept for safety)

  // this thread's slots of an operand tile [KT][TILE].  k-major operand: element e = tid + 256 i -> k = e / TILE,
  // column e % TILE (a wave reads 64 consecutive columns of one k-row).  k-minor operand: k = e % KT, column
  // e / KT (16 lanes read the 16 consecutive k of one column: a 64-byte piece of that row).
  constexpr int RS = 256 / TILE;                           // k-major: k rows between two slots of a thread
  constexpr int CS = 256 / KT;                             // k-minor: columns between two slots of a thread
  const int col = tid % TILE, row0 = tid / TILE;           // k-major slot
  const int kk = tid % KT, cc0 = tid / KT;                 // k-minor slot
  // Operand loads are BUFFER loads: (resource = wave-uniform base of the current tile, in SGPRs) + (per-thread
  // 32-bit byte offset) + (uniform offset, SGPR).  The loop carries no vector address arithmetic and no 64-bit
  // address temporaries — with per-thread pointers the register allocator recycled the staging registers as address
  // temporaries and had to wait for the loads in flight before issuing new ones.
  // Clamped columns: what lies outside M / N is loaded from the last valid column and never stored.
  uint32_t offA[AKM ? 1 : PER], offB[BKM ? 1 : PER];
  if (AKM) offA[0] = ((uint32_t)row0 * (uint32_t)lda + (uint32_t)min(m0 + col, M - 1)) * 4u;
  else {
#pragma unroll
    for (int i = 0; i < (AKM ? 1 : PER); ++i)
      offA[i] = ((uint32_t)min(m0 + cc0 + CS * i, M - 1) * (uint32_t)lda + (uint32_t)kk) * 4u;
  }
  if (BKM) offB[0] = ((uint32_t)row0 * (uint32_t)ldb + (uint32_t)min(n0 + col, N - 1)) * 4u;
  else {
#pragma unroll
    for (int i = 0; i < (BKM ? 1 : PER); ++i)
      offB[i] = ((uint32_t)min(n0 + cc0 + CS * i, N - 1) * (uint32_t)ldb + (uint32_t)kk) * 4u;
  }
  const char* uA = (const char*)(AKM ? A + (int64_t)kb * lda : A + kb);
  const char* uB = (const char*)(BKM ? B + (int64_t)kb * ldb : B + kb);
  const uint32_t stepA = (uint32_t)RS * (uint32_t)lda * 4u, stepB = (uint32_t)RS * (uint32_t)ldb * 4u;   // bytes
  auto bload = [](const char* base, uint32_t voff, uint32_t soff) -> float {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  };
  // Two register sets, two LDS buffers: while the matrix cores work on tile t (LDS), tile t + 1 waits in one
  // register set and the loads of tile t + 2 are issued into the other — every load has two tile-times to land.
  // A full tile is loaded unconditionally, so no select sits between a load and its LDS store and the wait lands
  // behind the MFMAs; only the last, partial tile predicates its rows (zero rows: exact no-ops).
  float ra0[PER], rb0[PER], ra1[PER], rb1[PER];
  auto gload = [&](float (&ra)[PER], float (&rb)[PER], int k0) {
    if (k0 + KT <= ke) {                                   // a full tile: no select between a load and its store
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        ra[i] = AKM ? bload(uA, offA[0], i * stepA) : bload(uA, offA[AKM ? 0 : i], 0);
        rb[i] = BKM ? bload(uB, offB[0], i * stepB) : bload(uB, offB[BKM ? 0 : i], 0);
      }
    } else {                                               // the last, partial tile: rows past the range are zeros
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const bool live_a = AKM ? k0 + row0 + RS * i < ke : k0 + kk < ke;
        const bool live_b = BKM ? k0 + row0 + RS * i < ke : k0 + kk < ke;
        ra[i] = live_a ? (AKM ? bload(uA, offA[0], i * stepA) : bload(uA, offA[AKM ? 0 : i], 0)) : 0.f;
        rb[i] = live_b ? (BKM ? bload(uB, offB[0], i * stepB) : bload(uB, offB[BKM ? 0 : i], 0)) : 0.f;
      }
    }
    uA += AKM ? (int64_t)KT * lda * 4 : (int64_t)KT * 4;
    uB += BKM ? (int64_t)KT * ldb * 4 : (int64_t)KT * 4;
  };
  auto sstore = [&](const float (&ra)[PER], const float (&rb)[PER], int buf) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if (AKM) sA[buf][row0 + RS * i][col] = ra[i];
      else sA[buf][kk][cc0 + CS * i] = ra[i];
      if (BKM) sB[buf][row0 + RS * i][col] = rb[i];
      else sB[buf][kk][cc0 + CS * i] = rb[i];
    }
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int s = 0; s < KT / 2; ++s) {
      float a[R], b[R];
#pragma unroll
      for (int x = 0; x < R; ++x) a[x] = sA[buf][2 * s + h][wm + 32 * x + j];
#pragma unroll
      for (int y = 0; y < R; ++y) b[y] = sB[buf][2 * s + h][wn + 32 * y + j];
#pragma unroll
      for (int x = 0; x < R; ++x)
#pragma unroll
        for (int y = 0; y < R; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x], b[y], acc[x][y], 0, 0, 0);
    }
  };
  gload(ra0, rb0, kb);
  sstore(ra0, rb0, 0);
  if (kb + KT < ke) gload(ra1, rb1, kb + KT);
  __syncthreads();
  for (int k0 = kb; k0 < ke; k0 += 2 * KT) {
    // even tile (LDS buffer 0); set 1 holds the next tile, set 0 receives the one after
    if (k0 + 2 * KT < ke) gload(ra0, rb0, k0 + 2 * KT);
    compute(0);
    if (k0 + KT < ke) sstore(ra1, rb1, 1);
    __syncthreads();
    if (k0 + KT >= ke) break;
    // odd tile (LDS buffer 1)
    if (k0 + 3 * KT < ke) gload(ra1, rb1, k0 + 3 * KT);
    compute(1);
    if (k0 + 2 * KT < ke) sstore(ra0, rb0, 0);
    __syncthreads();
  }
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int x = 0; x < R; ++x)
#pragma unroll
    for (int y = 0; y < R; ++y) {
      const int c = n0 + wn + 32 * y + j;
      const float bv = (bias_n && c < N) ? bias_n[c] : 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = m0 + wm + 32 * x + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (r < M && c < N) {
          float v = acc[x][y][reg];
          if (bias_n) v = v + bv;
          if (act >= 0) v = epilogue_act(act, v);
          C[(int64_t)r * ldc + c] = v;
        }
      }
    }
}

// C = (C if accumulate) + part[0] + part[1] + ... in split order
__global__ __launch_bounds__(256) void gemm_split_reduce_kernel(const float* __restrict__ parts, int splits,
                                                                int64_t split_stride, int M, int N, int64_t ldp,
                                                                float* __restrict__ C, int64_t ldc, int accumulate,
                                                                const float* __restrict__ bias_n, int act) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)M * N) return;
  const int m = (int)(e / N), n = (int)(e - (int64_t)m * N);
  float acc = accumulate ? C[(int64_t)m * ldc + n] : 0.f;
  const float* p = parts + (int64_t)m * ldp + n;
  int s = 0;
  for (; s + 8 <= splits; s += 8) {                      // eight loads in flight, added in split order
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = p[(int64_t)(s + q) * split_stride];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc = acc + v[q];
  }
  for (; s < splits; ++s) acc = acc + p[(int64_t)s * split_stride];
  if (bias_n) acc = acc + bias_n[n];
  if (act >= 0) acc = epilogue_act(act, acc);
  C[(int64_t)m * ldc + n] = acc;
}

// many splits (a tall-skinny product: 64 x 64 outputs over 70,839 rows): first the splits of one chunk of kRedChunk
// are added (grid.y = chunks, all of them side by side), then the chunk sums by gemm_split_reduce_kernel — the same
// left-to-right order inside a chunk and across the chunks, so the association is fixed (deterministic)
constexpr int kRedChunk = 32;
__global__ __launch_bounds__(256) void gemm_split_chunks_kernel(const float* __restrict__ parts, int splits,
                                                                int64_t split_stride, int64_t n_elems,
                                                                float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_elems) return;
  const int s0 = blockIdx.y * kRedChunk, s1 = min(splits, s0 + kRedChunk);
  const float* p = parts + e;
  float acc = 0.f;
  int s = s0;
  for (; s + 8 <= s1; s += 8) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = p[(int64_t)(s + q) * split_stride];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc = acc + v[q];
  }
  for (; s < s1; ++s) acc = acc + p[(int64_t)s * split_stride];
  out[(int64_t)blockIdx.y * n_elems + e] = acc;
}

// dst[c][r] = src[r][c]
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ src, int64_t ld_src, int rows,
                                                          int cols, float* __restrict__ dst, int64_t ld_dst) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  // (unconditional loads on a clamped position, masked afterwards: a load behind a branch is waited for before the
  //  next is issued)
  const int c = c0 + tx, cc = min(c, cols - 1);
  float val[16];
  if (rows > 0 && cols > 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) val[q] = src[(int64_t)min(r0 + q * 4 + ty, rows - 1) * ld_src + cc];
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) tile[q * 4 + ty][tx] = (rows > 0 && cols > 0 && r0 + q * 4 + ty < rows && c < cols) ? val[q] : 0.f;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int c = c0 + q * 4 + ty, r = r0 + tx;
    if (c < cols && r < rows) dst[(int64_t)c * ld_dst + r] = tile[tx][q * 4 + ty];
  }
}

}  // namespace

extern "C" {

int nrhip_gemm_workspace_bytes(int M, int N, int splits, size_t* bytes) {
  NR_REQUIRE(bytes && M >= 0 && N >= 0 && splits >= 1, NR_ERR_ARG, "gemm_workspace_bytes: bad arguments");
  const size_t chunks = splits > 2 * kRedChunk ? (size_t)(splits + kRedChunk - 1) / kRedChunk : 0;   // second level
  *bytes = splits > 1 ? ((size_t)splits + chunks) * (size_t)M * (size_t)N * sizeof(float) : 0;
  return NR_OK;
}

/* C[m][n] (+)= sum_k a(k, m) * b(k, n) (fp32 MFMA; see the file header), then + d_bias_n[n] (NULL: none) and the
 * activation `act` (-1 none, 0 tanh, 1 sigmoid, 2 relu, 3 identity).  a_kminor = 0: d_A is [K][lda] (k-major);
 * a_kminor = 1: d_A is [M][lda] with the contraction index contiguous; b_kminor likewise for d_B ([K][ldb] / [N][ldb]).
 * splits > 1: the contraction is cut into that many ranges computed side by side into d_ws (splits*M*N floats) and
 * added in order. */
int nrhip_gemm_f32(const float* d_A, int64_t lda, int a_kminor, const float* d_B, int64_t ldb, int b_kminor, int M,
                   int N, int K, float* d_C, int64_t ldc, int accumulate, const float* d_bias_n, int act, int splits,
                   void* d_ws, size_t ws_bytes, void* stream) {
  NR_REQUIRE(d_A && d_B && d_C && M >= 0 && N >= 0 && K >= 0 && lda >= (a_kminor ? K : M) &&
                 ldb >= (b_kminor ? K : N) && ldc >= N && lda < (1 << 24) && ldb < (1 << 24) && splits >= 1 &&
                 act >= -1 && act <= 3, NR_ERR_ARG, "gemm_f32: bad arguments");
  // a k-minor operand is addressed from its first row through one buffer resource (2 GB window, 32-bit offsets)
  NR_REQUIRE((!a_kminor || (int64_t)M * lda * 4 < (1ll << 31)) && (!b_kminor || (int64_t)N * ldb * 4 < (1ll << 31)),
             NR_ERR_UNSUPPORTED, "gemm_f32: a k-minor operand of %lld bytes (limit 2 GB: give it k-major)",
             (long long)((a_kminor ? (int64_t)M * lda : (int64_t)N * ldb) * 4));
  if (M == 0 || N == 0) return NR_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 block(256);
  if (K == 0) splits = 1;                                    // an empty contraction: C (+)= 0, then bias / activation
  // 128 x 128 block tiles when they fill the chip, 64 x 64 otherwise (more, smaller workgroups)
  // (a dimension that fits 64 would leave half of every 128-wide tile empty)
  const bool big = M > 64 && N > 64 && (int64_t)((M + 127) / 128) * ((N + 127) / 128) * splits >= 256;
  const int tile = big ? 128 : 64;
  const int tiles_m = (M + tile - 1) / tile, tiles_n = (N + tile - 1) / tile;
  const unsigned blocks = (unsigned)(((int64_t)tiles_m * tiles_n + 7) / 8 * 8);
  if (splits == 1 && K == 0) {
    // no k-steps: the kernel would return early; write the epilogue of a zero product through the reduce kernel
    hipLaunchKernelGGL(gemm_split_reduce_kernel, dim3((unsigned)(((int64_t)M * N + 255) / 256)), block, 0, st,
                       (const float*)d_C, 0, (int64_t)0, M, N, (int64_t)N, d_C, ldc, accumulate, d_bias_n, act);
    NR_LAUNCH_CHECK();
    return NR_OK;
  }
  int per = K, used = 1;
  float* out = d_C;
  int64_t ld_out = ldc, stride = 0;
  const float* bias = d_bias_n;
  int actv = act;
  bool acc = accumulate != 0;
  if (splits > 1) {
    const size_t chunks_ws = splits > 2 * kRedChunk ? (size_t)(splits + kRedChunk - 1) / kRedChunk : 0;
    NR_REQUIRE(d_ws && ws_bytes >= ((size_t)splits + chunks_ws) * M * N * sizeof(float), NR_ERR_WORKSPACE,
               "gemm_f32: workspace too small for %d splits (nrhip_gemm_workspace_bytes)", splits);
    per = (K + splits - 1) / splits;
    per = (per + KT - 1) / KT * KT;
    used = (K + per - 1) / per;
    out = (float*)d_ws;
    ld_out = N;
    stride = (int64_t)M * N;
    bias = nullptr;
    actv = -1;
    acc = false;
  }
  const dim3 grid(blocks, 1, (unsigned)used);
#define NR_GEMM_GO(TILE, ACCF, AK, BK)                                                                                 \
  hipLaunchKernelGGL((gemm_lds_kernel<TILE, ACCF, AK, BK>), grid, block, 0, st, d_A, lda, d_B, ldb, M, N, K, per, out, \
                     ld_out, stride, bias, actv, tiles_m, tiles_n)
#define NR_GEMM_LAYOUT(TILE, ACCF)                                                                \
  do {                                                                                            \
    if (!a_kminor && !b_kminor) NR_GEMM_GO(TILE, ACCF, true, true);                               \
    else if (a_kminor && !b_kminor) NR_GEMM_GO(TILE, ACCF, false, true);                          \
    else if (!a_kminor && b_kminor) NR_GEMM_GO(TILE, ACCF, true, false);                          \
    else NR_GEMM_GO(TILE, ACCF, false, false);                                                    \
  } while (0)
  if (big) { if (acc) NR_GEMM_LAYOUT(128, true); else NR_GEMM_LAYOUT(128, false); }
  else { if (acc) NR_GEMM_LAYOUT(64, true); else NR_GEMM_LAYOUT(64, false); }
#undef NR_GEMM_LAYOUT
#undef NR_GEMM_GO
  NR_LAUNCH_CHECK();
  if (splits > 1) {
    const float* parts = (const float*)d_ws;
    int n_parts = used;
    const unsigned eblocks = (unsigned)(((int64_t)M * N + 255) / 256);
    if (splits > 2 * kRedChunk) {                          // chunk sums first, side by side
      const int chunks = (used + kRedChunk - 1) / kRedChunk;
      float* lvl2 = (float*)d_ws + (size_t)splits * M * N;
      hipLaunchKernelGGL(gemm_split_chunks_kernel, dim3(eblocks, (unsigned)chunks), block, 0, st, parts, used,
                         (int64_t)M * N, (int64_t)M * N, lvl2);
      NR_LAUNCH_CHECK();
      parts = lvl2;
      n_parts = chunks;
    }
    hipLaunchKernelGGL(gemm_split_reduce_kernel, dim3(eblocks), block, 0, st, parts, n_parts, (int64_t)M * N, M, N,
                       (int64_t)N, d_C, ldc, accumulate, d_bias_n, act);
    NR_LAUNCH_CHECK();
  }
  return NR_OK;
}

/* the k-major / k-major form under its first name */
int nrhip_gemm_kmajor(const float* d_A, int64_t lda, const float* d_B, int64_t ldb, int M, int N, int K, float* d_C,
                      int64_t ldc, int accumulate, const float* d_bias_n, int act, int splits, void* d_ws,
                      size_t ws_bytes, void* stream) {
  return nrhip_gemm_f32(d_A, lda, 0, d_B, ldb, 0, M, N, K, d_C, ldc, accumulate, d_bias_n, act, splits, d_ws, ws_bytes,
                        stream);
}

int nrhip_transpose2d(const float* d_src, int64_t ld_src, int rows, int cols, float* d_dst, int64_t ld_dst,
                      void* stream) {
  NR_REQUIRE(d_src && d_dst && rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= rows, NR_ERR_ARG,
             "transpose2d: bad arguments");
  if (rows == 0 || cols == 0) return NR_OK;
  hipLaunchKernelGGL(transpose2d_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0,
                     (hipStream_t)stream, d_src, ld_src, rows, cols, d_dst, ld_dst);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // extern "C"
