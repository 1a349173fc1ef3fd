// gemm.hip — a general fp32 GEMM on the matrix cores for the dense products that are NOT the evaluation's
// scoring loop: the wide Mult-VAE shapes (conf/MultiVAE.properties:3 lists p_dim [200, 600]; MultiVAE.py:73-135:
// tf.matmul in q_graph / p_graph and their autodiff) whose hidden widths do not fit the register-resident 16/32-wide
// forms of vae.hip.
//
//     C[m][n] (+)= sum_k A[k][m] * B[k][n]          A: [K][lda], B: [K][ldb]  ("k-major": the contraction index is the
//                                                    slow one), C: [M][ldc] row-major
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

constexpr int kPairs = 8;        // k-pairs per unrolled step (32 loads in flight per lane)

// one wave = a 64 x 64 tile of C; block = 4 waves stacked along m
template <bool ACC>
__global__ __launch_bounds__(256) void gemm_kmajor_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, int M, int N, int K,
    int k_per_split, float* __restrict__ C, int64_t ldc, int64_t split_stride) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int m0 = (blockIdx.y * 4 + wave) * 64, n0 = blockIdx.x * 64;
  if (m0 >= M) return;
  const int kb = blockIdx.z * k_per_split, ke = min(K, kb + k_per_split);
  C += (int64_t)blockIdx.z * split_stride;
  // clamped operand columns (loads are unconditional; what lies outside M / N / K is multiplied by a zero)
  const int ma = min(m0 + j, M - 1), mb = min(m0 + 32 + j, M - 1);
  const int na = min(n0 + j, N - 1), nb = min(n0 + 32 + j, N - 1);
  f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
  if (ACC) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      const int ra = m0 + rr, rb = m0 + 32 + rr, ca = n0 + j, cb = n0 + 32 + j;
      if (ra < M && ca < N) c00[reg] = C[(int64_t)ra * ldc + ca];
      if (ra < M && cb < N) c01[reg] = C[(int64_t)ra * ldc + cb];
      if (rb < M && ca < N) c10[reg] = C[(int64_t)rb * ldc + ca];
      if (rb < M && cb < N) c11[reg] = C[(int64_t)rb * ldc + cb];
    }
  }
  for (int k0 = kb; k0 < ke; k0 += 2 * kPairs) {
    float a0[kPairs], a1[kPairs], b0[kPairs], b1[kPairs];
#pragma unroll
    for (int s = 0; s < kPairs; ++s) {
      const int k = k0 + 2 * s + h;
      const int kc = min(k, ke - 1);
      const float live = k < ke ? 1.f : 0.f;            // a zero A operand removes the step exactly (x + 0*b = x)
      a0[s] = A[(int64_t)kc * lda + ma] * live;
      a1[s] = A[(int64_t)kc * lda + mb] * live;
      b0[s] = B[(int64_t)kc * ldb + na];
      b1[s] = B[(int64_t)kc * ldb + nb];
    }
#pragma unroll
    for (int s = 0; s < kPairs; ++s) {
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], c11, 0, 0, 0);
    }
  }
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * h;
    const int ra = m0 + rr, rb = m0 + 32 + rr, ca = n0 + j, cb = n0 + 32 + j;
    if (ra < M && ca < N) C[(int64_t)ra * ldc + ca] = c00[reg];
    if (ra < M && cb < N) C[(int64_t)ra * ldc + cb] = c01[reg];
    if (rb < M && ca < N) C[(int64_t)rb * ldc + ca] = c10[reg];
    if (rb < M && cb < N) C[(int64_t)rb * ldc + cb] = c11[reg];
  }
}

// C = (C if accumulate) + part[0] + part[1] + ... in split order
__global__ __launch_bounds__(256) void gemm_split_reduce_kernel(const float* __restrict__ parts, int splits,
                                                                int64_t split_stride, int M, int N, int64_t ldp,
                                                                float* __restrict__ C, int64_t ldc, int accumulate) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)M * N) return;
  const int m = (int)(e / N), n = (int)(e - (int64_t)m * N);
  float acc = accumulate ? C[(int64_t)m * ldc + n] : 0.f;
  for (int s = 0; s < splits; ++s) acc = acc + parts[(int64_t)s * split_stride + (int64_t)m * ldp + n];
  C[(int64_t)m * ldc + n] = acc;
}

// dst[c][r] = src[r][c]
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ src, int64_t ld_src, int rows,
                                                          int cols, float* __restrict__ dst, int64_t ld_dst) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = r0 + q * 4 + ty, c = c0 + tx;
    tile[q * 4 + ty][tx] = (r < rows && c < cols) ? src[(int64_t)r * ld_src + c] : 0.f;
  }
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
  *bytes = splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 0;
  return NR_OK;
}

/* C[m][n] (+)= sum_k A[k][m] * B[k][n] (fp32 MFMA; see the file header).  splits > 1: the contraction is cut into
 * that many ranges computed side by side into d_ws (splits*M*N floats) and added in order. */
int nrhip_gemm_kmajor(const float* d_A, int64_t lda, const float* d_B, int64_t ldb, int M, int N, int K, float* d_C,
                      int64_t ldc, int accumulate, int splits, void* d_ws, size_t ws_bytes, void* stream) {
  NR_REQUIRE(d_A && d_B && d_C && M >= 0 && N >= 0 && K >= 0 && lda >= M && ldb >= N && ldc >= N && splits >= 1,
             NR_ERR_ARG, "gemm_kmajor: bad arguments");
  if (M == 0 || N == 0) return NR_OK;
  hipStream_t st = (hipStream_t)stream;
  if (K == 0) {
    if (!accumulate)
      for (int m = 0; m < M; ++m) NR_CHECK_HIP(hipMemsetAsync(d_C + (int64_t)m * ldc, 0, sizeof(float) * N, st));
    return NR_OK;
  }
  dim3 block(256);
  if (splits == 1) {
    dim3 grid((N + 63) / 64, (M + 255) / 256, 1);
    if (accumulate)
      hipLaunchKernelGGL(gemm_kmajor_kernel<true>, grid, block, 0, st, d_A, lda, d_B, ldb, M, N, K, K, d_C, ldc,
                         (int64_t)0);
    else
      hipLaunchKernelGGL(gemm_kmajor_kernel<false>, grid, block, 0, st, d_A, lda, d_B, ldb, M, N, K, K, d_C, ldc,
                         (int64_t)0);
    NR_LAUNCH_CHECK();
    return NR_OK;
  }
  NR_REQUIRE(d_ws && ws_bytes >= (size_t)splits * M * N * sizeof(float), NR_ERR_WORKSPACE,
             "gemm_kmajor: workspace too small for %d splits", splits);
  int per = (K + splits - 1) / splits;
  per = (per + 2 * kPairs - 1) / (2 * kPairs) * (2 * kPairs);
  const int used = (K + per - 1) / per;
  dim3 grid((N + 63) / 64, (M + 255) / 256, used);
  hipLaunchKernelGGL(gemm_kmajor_kernel<false>, grid, block, 0, st, d_A, lda, d_B, ldb, M, N, K, per, (float*)d_ws,
                     (int64_t)N, (int64_t)M * N);
  NR_LAUNCH_CHECK();
  hipLaunchKernelGGL(gemm_split_reduce_kernel, dim3((unsigned)(((int64_t)M * N + 255) / 256)), block, 0, st,
                     (const float*)d_ws, used, (int64_t)M * N, M, N, (int64_t)N, d_C, ldc, accumulate);
  NR_LAUNCH_CHECK();
  return NR_OK;
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
