// ngcf_wide.hip — NGCF propagation layers of ANY width (NGCF.py:31-33,271-286 take any embedding_size / layer_size;
// conf/NGCF.properties ships 16 / [16, 16], the NGCF paper uses 64 / [64, 64, 64]).  dense.hip is the fused,
// register-resident form for the shipped width; here a layer (NGCF.py:172-198)
//     T1 = S W_gc + b_gc            T2 = (E .* S) W_bi + b_bi                     S = A_hat E (SpMM)
//     Z  = leaky_relu(T1) + leaky_relu(T2)      E' = Z / keep * mask      out = E' / max(|E'|, 1e-6)  (l2_normalize)
// is strung by neurec_amd/ngcf_wide.py from the SpMM, the general fp32-MFMA GEMM (gemm.hip: both products, their
// weight gradients over the N rows, the transposed products of the backward pass) and the row-wise kernels below.
// One wave per node row, lanes over the columns (<= 256): the row norm and the dot with the incoming gradient are
// wave reductions.
#include "nr_common.h"
#include "neurec_hip.h"

namespace {

constexpr float kLeaky = 0.2f;
constexpr float kNormEps = 1e-12f;
constexpr int kMaxPer = 4;                                   // columns per lane: widths up to 256

__device__ __forceinline__ float lrelu(float x) { return x > 0.f ? x : __fmul_rn(x, kLeaky); }

// out[r][c] = a[r][c] * b[r][c]
__global__ __launch_bounds__(256) void ew_mul_kernel(const float* __restrict__ a, int64_t lda,
                                                     const float* __restrict__ b, int64_t ldb, int64_t rows, int cols,
                                                     float* __restrict__ out, int64_t ldo) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * cols) return;
  const int64_t r = e / cols;
  const int c = (int)(e - r * cols);
  out[r * ldo + c] = __fmul_rn(a[r * lda + c], b[r * ldb + c]);
}

__global__ __launch_bounds__(256) void ngcf_act_fwd_kernel(
    const float* __restrict__ T1, const float* __restrict__ T2, int64_t ldt, int64_t n_rows, int w, int w_pad,
    float keep, uint8_t* __restrict__ mask_io, int mask_given, uint64_t seed, uint64_t step, int layer,
    float* __restrict__ ego_out, int64_t lde, float* __restrict__ out, int64_t ldo) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= n_rows) return;
  const uint64_t key = nr::splitmix64(seed ^ (step * 0x9e3779b97f4a7c15ull + layer));
  float z[kMaxPer];
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < kMaxPer; ++q) {
    const int c = lane + 64 * q;
    z[q] = 0.f;
    if (c < w) {
      bool kp;
      if (mask_given) kp = mask_io[r * w + c] != 0;
      else {
        kp = (float)(nr::splitmix64(key ^ (uint64_t)(r * w + c)) >> 40) * (1.0f / 16777216.0f) < keep;
        mask_io[r * w + c] = kp ? 1 : 0;
      }
      const float zz = __fadd_rn(lrelu(T1[r * ldt + c]), lrelu(T2[r * ldt + c]));
      z[q] = kp ? zz / keep : 0.f;
      ss = fmaf(z[q], z[q], ss);
    }
  }
  ss = nr_wave_sum_f32(ss);
  const float inv = 1.0f / sqrtf(fmaxf(ss, kNormEps));
#pragma unroll
  for (int q = 0; q < kMaxPer; ++q) {
    const int c = lane + 64 * q;
    if (c < w_pad) ego_out[r * lde + c] = z[q];            // pad columns: zeros (the SpMM runs on padded rows)
    if (c < w) out[r * ldo + c] = __fmul_rn(z[q], inv);
  }
}

// dT1, dT2 from dLoss/d out_k (d_out), the gradient arriving through the next layer (d_ego_next, may be NULL),
// and the forward's E' (ego_next), T1, T2, mask
__global__ __launch_bounds__(256) void ngcf_act_bwd_kernel(
    const float* __restrict__ d_out, int64_t ldo, const float* __restrict__ d_ego_next, int64_t ldn,
    const float* __restrict__ ego_next, int64_t lde, const float* __restrict__ T1, const float* __restrict__ T2,
    int64_t ldt, const uint8_t* __restrict__ mask, int64_t n_rows, int w, float keep, float* __restrict__ dT1,
    float* __restrict__ dT2) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= n_rows) return;
  float z[kMaxPer], g[kMaxPer];
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < kMaxPer; ++q) {
    const int c = lane + 64 * q;
    z[q] = c < w ? ego_next[r * lde + c] : 0.f;
    g[q] = c < w ? d_out[r * ldo + c] : 0.f;
    ss = fmaf(z[q], z[q], ss);
  }
  ss = nr_wave_sum_f32(ss);
  const float inv = 1.0f / sqrtf(fmaxf(ss, kNormEps));
  float dot = 0.f;
#pragma unroll
  for (int q = 0; q < kMaxPer; ++q) dot = fmaf(g[q], __fmul_rn(z[q], inv), dot);
  dot = nr_wave_sum_f32(dot);
#pragma unroll
  for (int q = 0; q < kMaxPer; ++q) {
    const int c = lane + 64 * q;
    if (c >= w) continue;
    float v = ss > kNormEps ? (g[q] - (z[q] * inv) * dot) * inv : g[q] * inv;
    if (d_ego_next) v += d_ego_next[r * ldn + c];
    const float dz = mask[r * w + c] ? v / keep : 0.f;
    const float t1 = T1[r * ldt + c], t2 = T2[r * ldt + c];
    dT1[r * ldt + c] = t1 > 0.f ? dz : dz * kLeaky;
    dT2[r * ldt + c] = t2 > 0.f ? dz : dz * kLeaky;
  }
}

// dS = Y1 + Y2 .* ego ;  d_ego_direct = Y2 .* S      (Y1 = dT1 W_gc^T, Y2 = dT2 W_bi^T); pad columns: zeros
__global__ __launch_bounds__(256) void ngcf_mix_bwd_kernel(const float* __restrict__ Y1, const float* __restrict__ Y2,
                                                           int64_t ldy, const float* __restrict__ ego,
                                                           const float* __restrict__ S, int64_t lde, int64_t n_rows,
                                                           int w, int w_pad, float* __restrict__ dS,
                                                           float* __restrict__ d_ego_direct) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_rows * w_pad) return;
  const int64_t r = e / w_pad;
  const int c = (int)(e - r * w_pad);
  float ds = 0.f, de = 0.f;
  if (c < w) {
    const float y1 = Y1[r * ldy + c], y2 = Y2[r * ldy + c];
    ds = y1 + y2 * ego[r * lde + c];
    de = y2 * S[r * lde + c];
  }
  dS[r * lde + c] = ds;
  d_ego_direct[r * lde + c] = de;
}

// ---- alg_type = gcn / gcmc (NGCF.py:204-248) and node dropout (NGCF.py:334-362): the element-wise pieces -----------
// y = (flags & 1 ? leaky_relu(T) : T);  (flags & 2): y = mask ? y / keep : 0, the mask read (mask_given) or drawn from
// (seed, step, layer) with the hash ngcf_act_fwd uses.  out_a gets the row padded with zeros to w_pad (the next SpMM's
// operand), out_b the w real columns (a column block of the concatenated output); either may be NULL.
__global__ __launch_bounds__(256) void lrelu_drop_fwd_kernel(const float* __restrict__ T, int64_t ldt, int64_t n_rows,
                                                             int w, int w_pad, float keep, uint8_t* __restrict__ mask_io,
                                                             int mask_given, uint64_t seed, uint64_t step, int layer,
                                                             int flags, float* __restrict__ out_a, int64_t lda,
                                                             float* __restrict__ out_b, int64_t ldb) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_rows * w_pad) return;
  const int64_t r = e / w_pad;
  const int c = (int)(e - r * w_pad);
  float y = 0.f;
  if (c < w) {
    y = T[r * ldt + c];
    if (flags & 1) y = lrelu(y);
    if (flags & 2) {
      bool kp;
      if (mask_given) kp = mask_io[r * w + c] != 0;
      else {
        const uint64_t key = nr::splitmix64(seed ^ (step * 0x9e3779b97f4a7c15ull + layer));
        kp = (float)(nr::splitmix64(key ^ (uint64_t)(r * w + c)) >> 40) * (1.0f / 16777216.0f) < keep;
        mask_io[r * w + c] = kp ? 1 : 0;
      }
      y = kp ? y / keep : 0.f;
    }
    if (out_b) out_b[r * ldb + c] = y;
  }
  if (out_a) out_a[r * lda + c] = y;
}

// dT = (d_a + d_b) through the same ops backwards: (flags & 2) mask ? g / keep : 0, (flags & 1) times leaky_relu'(T)
__global__ __launch_bounds__(256) void lrelu_drop_bwd_kernel(const float* __restrict__ d_a, int64_t lda,
                                                             const float* __restrict__ d_b, int64_t ldb,
                                                             const float* __restrict__ T, int64_t ldt,
                                                             const uint8_t* __restrict__ mask, int64_t n_rows, int w,
                                                             float keep, int flags, float* __restrict__ dT) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_rows * w) return;
  const int64_t r = e / w;
  const int c = (int)(e - r * w);
  float g = d_a[r * lda + c];
  if (d_b) g += d_b[r * ldb + c];
  if (flags & 2) g = mask[r * w + c] ? g / keep : 0.f;
  if (flags & 1) g = T[r * ldt + c] > 0.f ? g : g * kLeaky;
  dT[r * (int64_t)w + c] = g;
}

// node dropout of the adjacency's stored entries: out[e] = keep_e ? vals[e] * (1 / keep) : 0 (NGCF.py:352-362:
// sparse_retain(X, floor(keep + uniform)) * (1 / keep)); keep_e read (given) or drawn from (seed, step) and written
__global__ __launch_bounds__(256) void edge_dropout_kernel(const float* __restrict__ vals, int64_t n, float keep,
                                                           uint8_t* __restrict__ keep_io, int given, uint64_t seed,
                                                           uint64_t step, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  bool kp;
  if (given) kp = keep_io[e] != 0;
  else {
    const uint64_t key = nr::splitmix64(seed ^ (step * 0x9e3779b97f4a7c15ull + 0x6e6f6465ull));
    kp = (float)(nr::splitmix64(key ^ (uint64_t)e) >> 40) * (1.0f / 16777216.0f) < keep;
    keep_io[e] = kp ? 1 : 0;
  }
  out[e] = kp ? __fmul_rn(vals[e], __fdiv_rn(1.0f, keep)) : 0.f;
}

__global__ __launch_bounds__(256) void gather_f32_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                         int64_t n, float* __restrict__ dst) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n) dst[e] = src[idx[e]];
}

}  // namespace

extern "C" {

int nrhip_ew_mul(const float* d_a, int64_t lda, const float* d_b, int64_t ldb, int64_t rows, int cols, float* d_out,
                 int64_t ldo, void* stream) {
  NR_REQUIRE(d_a && d_b && d_out && rows >= 0 && cols >= 1 && lda >= cols && ldb >= cols && ldo >= cols, NR_ERR_ARG,
             "ew_mul: bad arguments");
  if (rows == 0) return NR_OK;
  hipLaunchKernelGGL(ew_mul_kernel, dim3((unsigned)((rows * cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     d_a, lda, d_b, ldb, rows, cols, d_out, ldo);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

/* Z = leaky_relu(T1) + leaky_relu(T2); E' = Z / keep * mask -> d_ego_out [n_rows][lde] (columns w .. w_pad-1 zeroed);
 * l2_normalize(E') -> d_out [n_rows][ldo].  d_mask_io [n_rows][w] bytes: read when mask_given, else drawn from
 * (seed, step, layer) and written (NGCF.py:181-198). */
int nrhip_ngcf_act_fwd(const float* d_T1, const float* d_T2, int64_t ldt, int64_t n_rows, int w, int w_pad,
                       float keep, uint8_t* d_mask_io, int mask_given, uint64_t seed, uint64_t step, int layer,
                       float* d_ego_out, int64_t lde, float* d_out, int64_t ldo, void* stream) {
  NR_REQUIRE(d_T1 && d_T2 && d_mask_io && d_ego_out && d_out && n_rows >= 0 && w >= 1 && w <= 64 * kMaxPer &&
                 w_pad >= w && w_pad <= 64 * kMaxPer && ldt >= w && lde >= w_pad && ldo >= w && keep > 0.f &&
                 keep <= 1.f, NR_ERR_ARG, "ngcf_act_fwd: bad arguments (widths 1..256)");
  if (n_rows == 0) return NR_OK;
  hipLaunchKernelGGL(ngcf_act_fwd_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     d_T1, d_T2, ldt, n_rows, w, w_pad, keep, d_mask_io, mask_given, seed, step, layer, d_ego_out, lde,
                     d_out, ldo);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_ngcf_act_bwd(const float* d_dout, int64_t ldo, const float* d_dego_next, int64_t ldn,
                       const float* d_ego_next, int64_t lde, const float* d_T1, const float* d_T2, int64_t ldt,
                       const uint8_t* d_mask, int64_t n_rows, int w, float keep, float* d_dT1, float* d_dT2,
                       void* stream) {
  NR_REQUIRE(d_dout && d_ego_next && d_T1 && d_T2 && d_mask && d_dT1 && d_dT2 && n_rows >= 0 && w >= 1 &&
                 w <= 64 * kMaxPer && ldo >= w && lde >= w && ldt >= w && (!d_dego_next || ldn >= w) && keep > 0.f,
             NR_ERR_ARG, "ngcf_act_bwd: bad arguments (widths 1..256)");
  if (n_rows == 0) return NR_OK;
  hipLaunchKernelGGL(ngcf_act_bwd_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     d_dout, ldo, d_dego_next, ldn, d_ego_next, lde, d_T1, d_T2, ldt, d_mask, n_rows, w, keep, d_dT1,
                     d_dT2);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_lrelu_drop_fwd(const float* d_T, int64_t ldt, int64_t n_rows, int w, int w_pad, float keep, uint8_t* d_mask_io,
                         int mask_given, uint64_t seed, uint64_t step, int layer, int flags, float* d_out_a, int64_t lda,
                         float* d_out_b, int64_t ldb, void* stream) {
  NR_REQUIRE(d_T && (d_out_a || d_out_b) && n_rows >= 0 && w >= 1 && w_pad >= w && ldt >= w && (!d_out_a || lda >= w_pad) &&
                 (!d_out_b || ldb >= w) && (!(flags & 2) || (d_mask_io && keep > 0.f && keep <= 1.f)), NR_ERR_ARG,
             "lrelu_drop_fwd: bad arguments");
  if (n_rows == 0) return NR_OK;
  hipLaunchKernelGGL(lrelu_drop_fwd_kernel, dim3((unsigned)((n_rows * w_pad + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, d_T, ldt, n_rows, w, w_pad, keep, d_mask_io, mask_given, seed, step, layer, flags,
                     d_out_a, lda, d_out_b, ldb);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_lrelu_drop_bwd(const float* d_da, int64_t lda, const float* d_db, int64_t ldb, const float* d_T, int64_t ldt,
                         const uint8_t* d_mask, int64_t n_rows, int w, float keep, int flags, float* d_dT, void* stream) {
  NR_REQUIRE(d_da && d_dT && n_rows >= 0 && w >= 1 && lda >= w && (!d_db || ldb >= w) && (!(flags & 1) || (d_T && ldt >= w)) &&
                 (!(flags & 2) || (d_mask && keep > 0.f)), NR_ERR_ARG, "lrelu_drop_bwd: bad arguments");
  if (n_rows == 0) return NR_OK;
  hipLaunchKernelGGL(lrelu_drop_bwd_kernel, dim3((unsigned)((n_rows * w + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     d_da, lda, d_db, ldb, d_T, ldt, d_mask, n_rows, w, keep, flags, d_dT);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_edge_dropout(const float* d_vals, int64_t n, float keep, uint8_t* d_keep_io, int given, uint64_t seed,
                       uint64_t step, float* d_out, void* stream) {
  NR_REQUIRE(d_vals && d_keep_io && d_out && n >= 0 && keep > 0.f && keep <= 1.f, NR_ERR_ARG, "edge_dropout: bad arguments");
  if (n == 0) return NR_OK;
  hipLaunchKernelGGL(edge_dropout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_vals, n,
                     keep, d_keep_io, given, seed, step, d_out);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_gather_f32(const float* d_src, const int32_t* d_index, int64_t n, float* d_dst, void* stream) {
  NR_REQUIRE(d_src && d_index && d_dst && n >= 0, NR_ERR_ARG, "gather_f32: bad arguments");
  if (n == 0) return NR_OK;
  hipLaunchKernelGGL(gather_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_src,
                     d_index, n, d_dst);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_ngcf_mix_bwd(const float* d_Y1, const float* d_Y2, int64_t ldy, const float* d_ego, const float* d_S,
                       int64_t lde, int64_t n_rows, int w, int w_pad, float* d_dS, float* d_dego_direct,
                       void* stream) {
  NR_REQUIRE(d_Y1 && d_Y2 && d_ego && d_S && d_dS && d_dego_direct && n_rows >= 0 && w >= 1 && w_pad >= w &&
                 ldy >= w && lde >= w_pad, NR_ERR_ARG, "ngcf_mix_bwd: bad arguments");
  if (n_rows == 0) return NR_OK;
  hipLaunchKernelGGL(ngcf_mix_bwd_kernel, dim3((unsigned)((n_rows * w_pad + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, d_Y1, d_Y2, ldy, d_ego, d_S, lde, n_rows, w, w_pad, d_dS, d_dego_direct);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// The step as ONE native call: the launch sequence neurec_amd/ngcf_wide.py issued from Python.
// ---------------------------------------------------------------------------------------------------------------
static int ngcf_wide_check(const nrhip_ngcf_wide_buffers* b, const char* who) {
  NR_REQUIRE(b && b->plan && b->plan_t && b->E0p && b->Out && b->dOut, NR_ERR_ARG, "%s: null buffers", who);
  NR_REQUIRE(b->n_layers >= 0 && b->n_layers <= NRHIP_NGCF_WIDE_MAX_LAYERS, NR_ERR_UNSUPPORTED,
             "%s: %d layers (at most %d)", who, b->n_layers, NRHIP_NGCF_WIDE_MAX_LAYERS);
  NR_REQUIRE(b->dsum >= 1 && b->dsum <= 256, NR_ERR_UNSUPPORTED, "%s: concatenated width %d outside 1..256", who, b->dsum);
  return NR_OK;
}

extern "C" int nrhip_ngcf_wide_forward(const nrhip_ngcf_wide_buffers* b, int masks_given, uint64_t seed, uint64_t step,
                                       void* stream) {
  NR_TRY(ngcf_wide_check(b, "ngcf_wide_forward"));
  const int64_t N = b->n_nodes;
  NR_TRY(nrhip_copy2d(b->E0p, b->wp[0], b->Out, b->dsum, N, b->w[0], stream));
  for (int k = 0; k < b->n_layers; ++k) {
    const int wi = b->w[k], wo = b->w[k + 1], pi = b->wp[k], po = b->wp[k + 1];
    NR_TRY(nrhip_spmm_csr(b->plan, b->indptr, b->indices, b->vals, b->ego[k], pi, b->S[k], nullptr, nullptr, nullptr,
                          b->ws_fwd[k], b->ws_fwd_bytes[k], stream));
    NR_TRY(nrhip_ew_mul(b->ego[k], pi, b->S[k], pi, N, wi, b->X2[k], pi, stream));
    NR_TRY(nrhip_gemm_f32(b->S[k], pi, 1, b->W[k][0], wo, 0, (int)N, wo, wi, b->T1[k], wo, 0, b->W[k][1], -1, 1,
                          b->gemm_ws, 0, stream));
    NR_TRY(nrhip_gemm_f32(b->X2[k], pi, 1, b->W[k][2], wo, 0, (int)N, wo, wi, b->T2[k], wo, 0, b->W[k][3], -1, 1,
                          b->gemm_ws, 0, stream));
    NR_TRY(nrhip_ngcf_act_fwd(b->T1[k], b->T2[k], wo, N, wo, po, b->keep, b->mask[k], masks_given, seed, step, k,
                              b->ego[k + 1], po, b->Out + b->off[k + 1], b->dsum, stream));
  }
  return NR_OK;
}

extern "C" int nrhip_ngcf_wide_step(const nrhip_ngcf_wide_buffers* b, const int32_t* d_users, const int32_t* d_pos,
                                    const int32_t* d_neg, int batch, const uint64_t* d_plan, int masks_given,
                                    uint64_t seed, uint64_t step, float alpha, float beta1, float beta2, float eps,
                                    float* d_loss2, void* stream) {
  NR_TRY(ngcf_wide_check(b, "ngcf_wide_step"));
  NR_REQUIRE(d_users && d_pos && d_neg && batch >= 1 && batch <= b->max_batch, NR_ERR_ARG,
             "ngcf_wide_step: batch %d outside [1, %d]", batch, b->max_batch);
  const int64_t N = b->n_nodes;
  const int U = b->n_users, L = b->n_layers, dsum = b->dsum;
  NR_TRY(nrhip_ngcf_wide_forward(b, masks_given, seed, step, stream));
  NR_TRY(nrhip_lightgcn_mark_batch(d_users, d_pos, d_neg, batch, U, b->rows, b->flag, stream));
  NR_TRY(nrhip_bpr_mf_grad(b->Out, b->Out + (int64_t)U * dsum, dsum, U, d_users, d_pos, d_neg, batch, b->reg, b->dOut,
                           b->dOut + (int64_t)U * dsum, b->terms, d_loss2, d_plan, stream));
  const float* dego = nullptr;
  int dego_ld = 0;
  for (int k = L - 1; k >= 0; --k) {
    const int wi = b->w[k], wo = b->w[k + 1], pi = b->wp[k], po = b->wp[k + 1];
    NR_TRY(nrhip_ngcf_act_bwd(b->dOut + b->off[k + 1], dsum, dego, po, b->ego[k + 1], po, b->T1[k], b->T2[k], wo,
                              b->mask[k], N, wo, b->keep, b->dT1, b->dT2, stream));
    // weight gradients: contractions over the N rows, both operands k-major as stored
    NR_TRY(nrhip_gemm_f32(b->S[k], pi, 0, b->dT1, wo, 0, wi, wo, (int)N, b->gW[k][0], wo, 0, nullptr, -1, b->splits,
                          b->gemm_ws, b->gemm_ws_bytes, stream));
    NR_TRY(nrhip_gemm_f32(b->X2[k], pi, 0, b->dT2, wo, 0, wi, wo, (int)N, b->gW[k][2], wo, 0, nullptr, -1, b->splits,
                          b->gemm_ws, b->gemm_ws_bytes, stream));
    NR_TRY(nrhip_colsum_rows(b->dT1, wo, (int)N, wo, b->gW[k][1], b->cs_ws, b->cs_ws_bytes, stream));
    NR_TRY(nrhip_colsum_rows(b->dT2, wo, (int)N, wo, b->gW[k][3], b->cs_ws, b->cs_ws_bytes, stream));
    // Y1 = dT1 W_gc^T, Y2 = dT2 W_bi^T: both operands k-minor as stored
    NR_TRY(nrhip_gemm_f32(b->dT1, wo, 1, b->W[k][0], wo, 1, (int)N, wi, wo, b->Y1, wi, 0, nullptr, -1, 1, b->gemm_ws, 0,
                          stream));
    NR_TRY(nrhip_gemm_f32(b->dT2, wo, 1, b->W[k][2], wo, 1, (int)N, wi, wo, b->Y2, wi, 0, nullptr, -1, 1, b->gemm_ws, 0,
                          stream));
    NR_TRY(nrhip_ngcf_mix_bwd(b->Y1, b->Y2, wi, b->ego[k], b->S[k], pi, N, wi, pi, b->dS[k], b->dEd[k], stream));
    NR_TRY(nrhip_spmm_csr(b->plan_t, b->indptr_t, b->indices_t, b->vals_t, b->dS[k], pi, b->dEgo[k], b->dEd[k], nullptr,
                          nullptr, b->ws_bwd[k], b->ws_bwd_bytes[k], stream));       // dE_k = dBi .* S + A_hat^T dS
    dego = b->dEgo[k];
    dego_ld = pi;
  }
  const int w0 = b->w[0], p0 = b->wp[0];
  if (!dego) NR_TRY(nrhip_copy2d(b->dOut, dsum, b->gE0, p0, N, w0, stream));
  else NR_TRY(nrhip_add2d(b->dOut, dsum, dego, dego_ld, b->gE0, p0, N, w0, stream));
  float* vars[1 + 4 * NRHIP_NGCF_WIDE_MAX_LAYERS];
  float* ms[1 + 4 * NRHIP_NGCF_WIDE_MAX_LAYERS];
  float* vs[1 + 4 * NRHIP_NGCF_WIDE_MAX_LAYERS];
  float* gs[1 + 4 * NRHIP_NGCF_WIDE_MAX_LAYERS];
  int64_t sizes[1 + 4 * NRHIP_NGCF_WIDE_MAX_LAYERS];
  int32_t clear[1 + 4 * NRHIP_NGCF_WIDE_MAX_LAYERS];
  int n = 0;
  vars[n] = b->E0p; ms[n] = b->mE; vs[n] = b->vE; gs[n] = b->gE0; sizes[n] = N * p0; clear[n] = 0; ++n;
  for (int k = 0; k < L; ++k)
    for (int j = 0; j < 4; ++j) {
      vars[n] = b->W[k][j]; ms[n] = b->mW[k][j]; vs[n] = b->vW[k][j]; gs[n] = b->gW[k][j];
      sizes[n] = (j % 2) ? b->w[k + 1] : (int64_t)b->w[k] * b->w[k + 1];
      clear[n] = 0;
      ++n;
    }
  for (int lo = 0; lo < n; lo += 32) {
    const int m = n - lo < 32 ? n - lo : 32;
    NR_TRY(nrhip_adam_dense_tf_multi(m, vars + lo, ms + lo, vs + lo, gs + lo, sizes + lo, clear + lo, alpha, beta1, beta2,
                                     eps, stream));
  }
  return nrhip_rows_clear(b->rows, 3 * batch, dsum, b->dOut, nullptr, nullptr, nullptr, b->flag, stream);
}
