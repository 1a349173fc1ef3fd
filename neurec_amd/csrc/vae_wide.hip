// vae_wide.hip — Mult-VAE for ANY p_dim (conf/MultiVAE.properties:3 lists [200, 600] and [200] next to the shipped
// [16, 32]; MultiVAE.py:46-135 builds q_dims = reversed(p_dim + [I]) layers of arbitrary number and width).
// vae.hip is the register-resident form for the shipped two-layer shape with widths <= 32; here every piece is
// width-generic and the three large products (logits, dW of the last decoder layer, d(hidden) through it) go to the
// matrix cores through gemm.hip.  Same arithmetic definitions as vae.hip / oracle.train.multivae_general:
//   first encoder layer   a = bag-sum over the user's train items of (1/sqrt(n_u) / keep * mask) * W_q0[item]  (+ b)
//                         — the multi-hot row is never densified (the reference fills [B][I] on the host: :152-165)
//   dense layers          y = act(x W + b), k-ascending fmaf chain per output (gemm.hip, bias + activation epilogue)
//   sampling              z = mu + is_training * eps * exp(logvar / 2); KL_b = 1/2 sum(-logvar + exp(logvar) + mu^2 - 1)
//   decoder loss          nll_b = -sum_{i in items(b)} log_softmax(logits_b)_i;  dlogits = (softmax * n_b - x) / B
#include "nr_common.h"

namespace {

enum { ACT_TANH = 0, ACT_SIGMOID = 1, ACT_RELU = 2, ACT_IDENTITY = 3 };

__device__ __forceinline__ float act_fwd(int a, float x) {
  if (a == ACT_TANH) return tanhf(x);
  if (a == ACT_SIGMOID) return 1.0f / (1.0f + expf(-x));
  if (a == ACT_RELU) return fmaxf(x, 0.f);
  return x;
}
__device__ __forceinline__ float act_bwd(int a, float y) {       // d/d(pre-activation), from the output y
  if (a == ACT_TANH) return 1.0f - y * y;
  if (a == ACT_SIGMOID) return y * (1.0f - y);
  if (a == ACT_RELU) return y > 0.f ? 1.0f : 0.f;
  return 1.0f;
}
__device__ __forceinline__ float uniform01(uint64_t h) {
  return ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
}

// first encoder layer from the CSR row: one wave per (batch row, 64-column chunk of the output), a block = 4
// neighbouring chunks of one row; the row's (item, value) pairs are dealt 64 at a time and walked 8 gathers at a time
__global__ __launch_bounds__(256) void vae_bag_fwd_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const int32_t* __restrict__ rows,
    int batch, int width, const float* __restrict__ W, const float* __restrict__ bias, int act, float keep,
    const float* __restrict__ drop_given, uint64_t seed, uint64_t step, float* __restrict__ h0val,
    float* __restrict__ Y) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.y;
  const int c0 = (blockIdx.x * 4 + wave) * NR_WAVE;
  if (c0 >= width) return;
  const int c = c0 + lane, cc = min(c, width - 1);
  const int64_t u = rows[r];
  const int64_t b = indptr[u], e = indptr[u + 1];
  const int n = (int)(e - b);
  const float inv = 1.0f / sqrtf(fmaxf((float)n, 1e-12f));      // l2_normalize of a 0/1 row
  const uint64_t drop_key = nr::splitmix64(seed ^ (step * 0x9e3779b97f4a7c15ull));
  float acc = 0.f;
  for (int64_t t0 = b; t0 < e; t0 += NR_WAVE) {
    const int nn = (int)min((int64_t)NR_WAVE, e - t0);
    int my_item = 0;
    float my_val = 0.f;                                      // lanes past the row: exact no-ops below (acc + 0 * w)
    if (lane < nn) {
      const int64_t t = t0 + lane;
      float kp;
      if (drop_given) kp = drop_given[t];
      else kp = (keep >= 1.0f || uniform01(nr::splitmix64(drop_key ^ (uint64_t)t)) < keep) ? 1.f : 0.f;
      my_val = (inv / keep) * kp;                           // x / keep_prob * mask (tf.nn.dropout)
      my_item = indices[t];
      if (c0 == 0 && h0val) h0val[t] = my_val;
    }
    for (int s0 = 0; s0 < nn; s0 += 8) {                    // ascending item order, 8 row gathers in flight
      float w[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int item = __shfl(my_item, min(s0 + q, nn - 1), NR_WAVE);
        w[q] = W[(int64_t)item * width + cc];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float v = __shfl(my_val, (s0 + q) & 63, NR_WAVE);
        acc = fmaf(v, w[q], acc);
      }
    }
  }
  if (c < width) {
    acc = acc + bias[c];
    Y[(int64_t)r * width + c] = act >= 0 ? act_fwd(act, acc) : acc;
  }
}

// dA[b][j] = dY[b][j] * act'(Y[b][j])
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                      int64_t n, int act, float* __restrict__ dA) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dA[i] = dY[i] * act_bwd(act, Y[i]);
}

// H2 [B][2z] = [mu | logvar]  ->  ZS, EPSSTD, KLb
__global__ __launch_bounds__(256) void vae_sample_kernel(const float* __restrict__ H2, int batch, int z,
                                                         const float* __restrict__ eps_given, float is_training,
                                                         uint64_t seed, uint64_t step, float* __restrict__ EPSSTD,
                                                         float* __restrict__ ZS, float* __restrict__ KLb) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= batch) return;
  float kl = 0.f;
  for (int c = lane; c < z; c += NR_WAVE) {
    const float mu = H2[(int64_t)r * 2 * z + c], logvar = H2[(int64_t)r * 2 * z + z + c];
    const float sd = expf(0.5f * logvar);
    float eps;
    if (eps_given) eps = eps_given[(int64_t)r * z + c];
    else {                                                  // Box-Muller, N(0, 0.01^2): MultiVAE.py:111
      const uint64_t k0 = nr::splitmix64(nr::splitmix64(seed ^ 0xabcdull ^ (step << 20)) ^ ((uint64_t)r * z + c));
      const float u1 = uniform01(k0), u2 = uniform01(nr::splitmix64(k0));
      eps = 0.01f * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
    }
    const float es = eps * sd;
    EPSSTD[(int64_t)r * z + c] = es;
    ZS[(int64_t)r * z + c] = mu + is_training * es;
    kl += 0.5f * (-logvar + expf(logvar) + mu * mu - 1.0f);
  }
  kl = nr_wave_sum_f32(kl);
  if (lane == 0) KLb[r] = kl;
}

// dH2 = [dmu | dlogvar] from dZ (MultiVAE.py:86,111-113 differentiated; batch mean of the KL term)
__global__ __launch_bounds__(256) void vae_sample_bwd_kernel(const float* __restrict__ dZ,
                                                             const float* __restrict__ H2,
                                                             const float* __restrict__ EPSSTD, int batch, int z,
                                                             float anneal, float* __restrict__ dH2) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= (int64_t)batch * z) return;
  const int r = (int)(o / z), c = (int)(o - (int64_t)r * z);
  const float mu = H2[(int64_t)r * 2 * z + c], logvar = H2[(int64_t)r * 2 * z + z + c];
  const float dz = dZ[o], invB = 1.0f / (float)batch;
  dH2[(int64_t)r * 2 * z + c] = dz + anneal * mu * invB;
  dH2[(int64_t)r * 2 * z + z + c] = dz * EPSSTD[o] * 0.5f + anneal * 0.5f * (expf(logvar) - 1.0f) * invB;
}

// one workgroup per batch row: log-sum-exp, nll of the row's items, then dLoss/dlogits in place
__global__ __launch_bounds__(1024) void vae_softmax_dlogits_kernel(float* __restrict__ S, int64_t ld, int cols,
                                                                  const int64_t* __restrict__ indptr,
                                                                  const int32_t* __restrict__ indices,
                                                                  const int32_t* __restrict__ rows, int batch,
                                                                  float* __restrict__ nll) {
  __shared__ float s_red[1024];
  const int r = blockIdx.x, tid = threadIdx.x;
  float* row = S + (int64_t)r * ld;
  float mx = -INFINITY;
  for (int i = tid; i < cols; i += 1024) mx = fmaxf(mx, row[i]);
  s_red[tid] = mx;
  __syncthreads();
  for (int s = 512; s >= 1; s >>= 1) { if (tid < s) s_red[tid] = fmaxf(s_red[tid], s_red[tid + s]); __syncthreads(); }
  mx = s_red[0];
  __syncthreads();
  float sum = 0.f;
  for (int i = tid; i < cols; i += 1024) sum += expf(row[i] - mx);
  s_red[tid] = sum;
  __syncthreads();
  for (int s = 512; s >= 1; s >>= 1) { if (tid < s) s_red[tid] += s_red[tid + s]; __syncthreads(); }
  const float lse = mx + logf(s_red[0]);
  __syncthreads();
  const int64_t u = rows[r];
  const int64_t b = indptr[u], e = indptr[u + 1];
  float ll = 0.f;
  for (int64_t t = b + tid; t < e; t += 1024) ll += row[indices[t]] - lse;
  s_red[tid] = ll;
  __syncthreads();
  for (int s = 512; s >= 1; s >>= 1) { if (tid < s) s_red[tid] += s_red[tid + s]; __syncthreads(); }
  if (tid == 0) nll[r] = -s_red[0];
  __syncthreads();
  const float nb = (float)(e - b), invB = 1.0f / (float)batch;
  for (int i = tid; i < cols; i += 1024) row[i] = expf(row[i] - lse) * nb * invB;
  __syncthreads();
  for (int64_t t = b + tid; t < e; t += 1024) row[indices[t]] -= invB;       // distinct items: no conflict
}

// dW_q0[item][:] += h0val * dA1[b][:] over the batch's CSR entries (fp32 atomics, as vae.hip's scatter); one wave per
// (batch row, 64-column chunk): the atomics of one item row are 256-byte segments, fire-and-forget
__global__ __launch_bounds__(256) void vae_dwq0_wide_kernel(const int64_t* __restrict__ indptr,
                                                            const int32_t* __restrict__ indices,
                                                            const int32_t* __restrict__ rows, int batch, int width,
                                                            const float* __restrict__ h0val,
                                                            const float* __restrict__ DA1,
                                                            float* __restrict__ dWq0) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.y;
  const int c = (blockIdx.x * 4 + wave) * NR_WAVE + lane;
  if (c >= width) return;
  const int64_t u = rows[r];
  const int64_t b = indptr[u], e = indptr[u + 1];
  const float g = DA1[(int64_t)r * width + c];
  for (int64_t t = b; t < e; ++t) atomicAdd(&dWq0[(int64_t)indices[t] * width + c], h0val[t] * g);
}

}  // namespace

extern "C" {

int nrhip_vae_bag_fwd(const int64_t* d_indptr, const int32_t* d_indices, const int32_t* d_rows, int batch,
                      int width, const float* d_W, const float* d_bias, int act, float keep,
                      const float* d_drop_given, uint64_t seed, uint64_t step, float* d_h0val, float* d_Y,
                      void* stream) {
  NR_REQUIRE(d_indptr && d_indices && d_rows && d_W && d_bias && d_Y && batch >= 0 && width >= 1 && act >= -1 &&
                 act <= 3 && keep > 0.f && keep <= 1.f, NR_ERR_ARG, "vae_bag_fwd: bad arguments");
  if (batch == 0) return NR_OK;
  hipLaunchKernelGGL(vae_bag_fwd_kernel, dim3((width + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, d_indptr,
                     d_indices, d_rows, batch, width, d_W, d_bias, act, keep, d_drop_given, seed, step, d_h0val, d_Y);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_act_bwd(const float* d_dY, const float* d_Y, int64_t n, int act, float* d_dA, void* stream) {
  NR_REQUIRE(d_dY && d_Y && d_dA && n >= 0 && act >= 0 && act <= 3, NR_ERR_ARG, "act_bwd: bad arguments");
  if (n == 0) return NR_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_dY, d_Y,
                     n, act, d_dA);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_vae_sample(const float* d_H2, int batch, int z, const float* d_eps_given, float is_training, uint64_t seed,
                     uint64_t step, float* d_EPSSTD, float* d_ZS, float* d_KLb, void* stream) {
  NR_REQUIRE(d_H2 && d_EPSSTD && d_ZS && d_KLb && batch >= 0 && z >= 1, NR_ERR_ARG, "vae_sample: bad arguments");
  if (batch == 0) return NR_OK;
  hipLaunchKernelGGL(vae_sample_kernel, dim3((batch + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_H2, batch, z,
                     d_eps_given, is_training, seed, step, d_EPSSTD, d_ZS, d_KLb);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_vae_sample_bwd(const float* d_dZ, const float* d_H2, const float* d_EPSSTD, int batch, int z, float anneal,
                         float* d_dH2, void* stream) {
  NR_REQUIRE(d_dZ && d_H2 && d_EPSSTD && d_dH2 && batch >= 0 && z >= 1, NR_ERR_ARG, "vae_sample_bwd: bad arguments");
  if (batch == 0) return NR_OK;
  hipLaunchKernelGGL(vae_sample_bwd_kernel, dim3((unsigned)(((int64_t)batch * z + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, d_dZ, d_H2, d_EPSSTD, batch, z, anneal, d_dH2);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* logits slab S[batch][ld] (cols valid) -> d_nll[b] and, IN PLACE, dLoss/dlogits = (softmax * n_b - x) / batch */
int nrhip_vae_softmax_dlogits(float* d_S, int64_t ld, int batch, int cols, const int64_t* d_indptr,
                              const int32_t* d_indices, const int32_t* d_rows, float* d_nll, void* stream) {
  NR_REQUIRE(d_S && d_indptr && d_indices && d_rows && d_nll && batch >= 0 && cols >= 1 && ld >= cols, NR_ERR_ARG,
             "vae_softmax_dlogits: bad arguments");
  if (batch == 0) return NR_OK;
  hipLaunchKernelGGL(vae_softmax_dlogits_kernel, dim3(batch), dim3(1024), 0, (hipStream_t)stream, d_S, ld, cols,
                     d_indptr, d_indices, d_rows, batch, d_nll);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_vae_dwq0_wide(const int64_t* d_indptr, const int32_t* d_indices, const int32_t* d_rows, int batch,
                        int width, const float* d_h0val, const float* d_DA1, float* d_dWq0, void* stream) {
  NR_REQUIRE(d_indptr && d_indices && d_rows && d_h0val && d_DA1 && d_dWq0 && batch >= 0 && width >= 1, NR_ERR_ARG,
             "vae_dwq0_wide: bad arguments");
  if (batch == 0) return NR_OK;
  hipLaunchKernelGGL(vae_dwq0_wide_kernel, dim3((width + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, d_indptr,
                     d_indices, d_rows, batch, width, d_h0val, d_DA1, d_dWq0);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // extern "C"

namespace {
// out[chunk][c] = sum over the chunk's rows of X[r][c]: a block = 64 columns x 16 row groups; group g adds rows
// g, g + 16, ... of the chunk in order, then the 16 partial sums are added in group order (a fixed association:
// deterministic).  Tall inputs (NGCF's [N][w] gradients: 70,839 rows) are cut into chunks of rows_per_chunk rows —
// grid.y — whose sums a second launch of the same kernel adds in chunk order.
__global__ __launch_bounds__(1024) void colsum_rows_kernel(const float* __restrict__ X, int64_t ld, int rows, int cols,
                                                           int rows_per_chunk, float* __restrict__ out) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, cc = min(c, cols - 1);
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float acc = 0.f;
  for (int r = r0 + g; r < r1; r += 16) acc = acc + X[(int64_t)r * ld + cc];
  part[g][lane] = acc;
  __syncthreads();
  if (g == 0 && c < cols) {
    float t = part[0][lane];
#pragma unroll
    for (int q = 1; q < 16; ++q) t = t + part[q][lane];
    out[(int64_t)blockIdx.y * cols + c] = t;
  }
}
}  // namespace

/* d_out[c] = sum_r d_X[r][c] (bias gradients).  rows > 2048 needs d_ws of ceil(rows / 512) * cols floats. */
extern "C" int nrhip_colsum_rows(const float* d_X, int64_t ld, int rows, int cols, float* d_out, void* d_ws,
                                 size_t ws_bytes, void* stream) {
  NR_REQUIRE(d_X && d_out && rows >= 0 && cols >= 1 && ld >= cols, NR_ERR_ARG, "colsum_rows: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const unsigned bx = (unsigned)((cols + 63) / 64);
  if (rows <= 2048) {
    hipLaunchKernelGGL(colsum_rows_kernel, dim3(bx, 1), dim3(1024), 0, st, d_X, ld, rows, cols, rows > 0 ? rows : 1,
                       d_out);
    NR_LAUNCH_CHECK();
    return NR_OK;
  }
  const int per = 512, chunks = (rows + per - 1) / per;
  NR_REQUIRE(d_ws && ws_bytes >= (size_t)chunks * cols * sizeof(float), NR_ERR_WORKSPACE,
             "colsum_rows: workspace %zu < %zu", ws_bytes, (size_t)chunks * cols * sizeof(float));
  NR_REQUIRE(chunks <= 65535, NR_ERR_UNSUPPORTED, "colsum_rows: %d rows", rows);
  hipLaunchKernelGGL(colsum_rows_kernel, dim3(bx, chunks), dim3(1024), 0, st, d_X, ld, rows, cols, per, (float*)d_ws);
  NR_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_rows_kernel, dim3(bx, 1), dim3(1024), 0, st, (const float*)d_ws, (int64_t)cols, chunks,
                     cols, chunks, d_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}
