// dense.hip — the dense half of an NGCF propagation layer (second-model coverage, config 5).
//
// Stands in for the TF ops of NGCF._create_ngcf_embed (model/general_recommender/NGCF.py:181-198)
// and their autodiff.  Per layer, with S = Â·E from the SpMM kernel (NGCF.py:174-179):
//     T1 = S·W_gc + b_gc            T2 = (E ⊙ S)·W_bi + b_bi
//     Z  = leaky_relu(T1) + leaky_relu(T2)                  (alpha = 0.2)
//     E' = dropout(Z, keep) = Z / keep · mask               (always on, also at evaluation: NGCF.py:193)
//     out_k = l2_normalize(E', axis=1)                       (x · rsqrt(max(Σx², 1e-12)))
// The layer width is 16 (conf/NGCF.properties: embedding_size=16, layer_size=[16,16]); a row is
// 64 bytes, so ONE THREAD owns one node row and keeps it in registers, the two 16×16 weight
// matrices sit in LDS, and the row-wise normalisation needs no cross-lane traffic.  Everything
// here is a streaming pass over [N][16] buffers (4.5 MB at gowalla): HBM/launch-bound, fused so
// that a layer is one forward kernel and (backward) one row kernel + one weight-gradient kernel.
//
// Weight gradients dW = Xᵀ·G (X, G: [N][16]) are tall-skinny products reduced over the N rows:
// they run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, one wave per 256-row slab) and are
// combined over slabs in slab order (deterministic).
#include "nr_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRowsPerBlock = 256;
constexpr float kLeaky = 0.2f;
constexpr float kNormEps = 1e-12f;

__device__ __forceinline__ float lrelu(float x) { return x > 0.f ? x : __fmul_rn(x, kLeaky); }

// dropout draw for element `e` of layer `layer` at step `step`: keep with probability `keep`
__device__ __forceinline__ bool keep_draw(uint64_t seed, uint64_t step, int layer, uint64_t e,
                                          float keep) {
  const uint64_t h = nr::splitmix64(nr::splitmix64(seed ^ (step * 0x9e3779b97f4a7c15ull + layer)) ^ e);
  return (float)(h >> 40) * (1.0f / 16777216.0f) < keep;
}

template <int D>
struct RowMath {
  // T = x·W + b with W row-major [D][D] in LDS: k-ascending fused chain per output column
  __device__ static __forceinline__ void affine(const float (&x)[D], const float* W, const float* b,
                                                float (&t)[D]) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float acc = b[j];
#pragma unroll
      for (int k = 0; k < D; ++k) acc = fmaf(x[k], W[k * D + j], acc);
      t[j] = acc;
    }
  }
  // y = g·Wᵀ
  __device__ static __forceinline__ void affine_t(const float (&g)[D], const float* W, float (&y)[D]) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) acc = fmaf(g[j], W[k * D + j], acc);
      y[k] = acc;
    }
  }
};

template <int D>
__device__ __forceinline__ void load_row(const float* __restrict__ p, int64_t row, int64_t ld,
                                         float (&x)[D]) {
#pragma unroll
  for (int c = 0; c < D; c += 4) {
    const float4 v = *reinterpret_cast<const float4*>(p + row * ld + c);
    x[c] = v.x; x[c + 1] = v.y; x[c + 2] = v.z; x[c + 3] = v.w;
  }
}
template <int D>
__device__ __forceinline__ void store_row(float* __restrict__ p, int64_t row, int64_t ld,
                                          const float (&x)[D]) {
#pragma unroll
  for (int c = 0; c < D; c += 4)
    *reinterpret_cast<float4*>(p + row * ld + c) = make_float4(x[c], x[c + 1], x[c + 2], x[c + 3]);
}

template <int D>
__device__ __forceinline__ void stage_weights(const float* Wg, const float* bg, const float* Wb,
                                              const float* bb, float* s) {
  for (int i = threadIdx.x; i < D * D; i += blockDim.x) { s[i] = Wg[i]; s[D * D + i] = Wb[i]; }
  for (int i = threadIdx.x; i < D; i += blockDim.x) { s[2 * D * D + i] = bg[i]; s[2 * D * D + D + i] = bb[i]; }
  __syncthreads();
}

// forward of one layer; mask_io: if mask_given it is read, else it is drawn and written
template <int D>
__global__ __launch_bounds__(kRowsPerBlock) void ngcf_layer_fwd_kernel(
    const float* __restrict__ ego, const float* __restrict__ S, const float* __restrict__ Wg,
    const float* __restrict__ bg, const float* __restrict__ Wb, const float* __restrict__ bb,
    int64_t n_rows, float keep, uint8_t* __restrict__ mask_io, int mask_given, uint64_t seed,
    uint64_t step, int layer, float* __restrict__ ego_out, float* __restrict__ out, int64_t ldo) {
  __shared__ float s_w[2 * D * D + 2 * D];
  stage_weights<D>(Wg, bg, Wb, bb, s_w);
  const int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.x;
  if (r >= n_rows) return;
  float e[D], s[D], bi[D], t1[D], t2[D], z[D];
  load_row<D>(ego, r, D, e);
  load_row<D>(S, r, D, s);
#pragma unroll
  for (int k = 0; k < D; ++k) bi[k] = __fmul_rn(e[k], s[k]);
  RowMath<D>::affine(s, s_w, s_w + 2 * D * D, t1);
  RowMath<D>::affine(bi, s_w + D * D, s_w + 2 * D * D + D, t2);
  float ss = 0.f;
  // the row's D mask bytes travel as 16-byte words; a fresh draw takes four 16-bit uniforms from
  // each 64-bit hash (D/4 hashes per row)
  static_assert(D % 16 == 0, "mask rows are moved as uint4");
  uint32_t mw[D / 4];
  if (mask_given) {
#pragma unroll
    for (int q = 0; q < D / 16; ++q) {
      const uint4 m4 = reinterpret_cast<const uint4*>(mask_io + r * D)[q];
      mw[4 * q] = m4.x; mw[4 * q + 1] = m4.y; mw[4 * q + 2] = m4.z; mw[4 * q + 3] = m4.w;
    }
  } else {
    const uint64_t key = nr::splitmix64(seed ^ (step * 0x9e3779b97f4a7c15ull + layer));
#pragma unroll
    for (int q = 0; q < D / 4; ++q) {
      const uint64_t hsh = nr::splitmix64(key ^ ((uint64_t)r * (D / 4) + q));
      uint32_t w = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if ((float)((hsh >> (16 * i)) & 0xffffu) * (1.0f / 65536.0f) < keep) w |= 1u << (8 * i);
      mw[q] = w;
    }
#pragma unroll
    for (int q = 0; q < D / 16; ++q)
      reinterpret_cast<uint4*>(mask_io + r * D)[q] =
          make_uint4(mw[4 * q], mw[4 * q + 1], mw[4 * q + 2], mw[4 * q + 3]);
  }
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const bool kp = ((mw[k / 4] >> (8 * (k % 4))) & 0xffu) != 0;
    const float zz = __fadd_rn(lrelu(t1[k]), lrelu(t2[k]));
    z[k] = kp ? zz / keep : 0.f;
    ss = fmaf(z[k], z[k], ss);
  }
  const float inv = 1.0f / sqrtf(fmaxf(ss, kNormEps));
  store_row<D>(ego_out, r, D, z);
#pragma unroll
  for (int k = 0; k < D; ++k) z[k] = __fmul_rn(z[k], inv);
  store_row<D>(out, r, ldo, z);
}

// backward of one layer (row part): recomputes the forward values of the row from (ego, S, mask)
template <int D>
__global__ __launch_bounds__(kRowsPerBlock) void ngcf_layer_bwd_kernel(
    const float* __restrict__ ego, const float* __restrict__ S, const float* __restrict__ Wg,
    const float* __restrict__ bg, const float* __restrict__ Wb, const float* __restrict__ bb,
    int64_t n_rows, float keep, const uint8_t* __restrict__ mask, const float* __restrict__ d_out,
    int64_t ldo, const float* __restrict__ d_ego_next, float* __restrict__ dS,
    float* __restrict__ d_ego_direct, float* __restrict__ dT1, float* __restrict__ dT2) {
  __shared__ float s_w[2 * D * D + 2 * D];
  stage_weights<D>(Wg, bg, Wb, bb, s_w);
  const int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.x;
  if (r >= n_rows) return;
  float e[D], s[D], bi[D], t1[D], t2[D], z[D], g[D];
  load_row<D>(ego, r, D, e);
  load_row<D>(S, r, D, s);
#pragma unroll
  for (int k = 0; k < D; ++k) bi[k] = __fmul_rn(e[k], s[k]);
  RowMath<D>::affine(s, s_w, s_w + 2 * D * D, t1);
  RowMath<D>::affine(bi, s_w + D * D, s_w + 2 * D * D + D, t2);
  float ss = 0.f;
  bool kp[D];
  uint32_t mw[D / 4];
#pragma unroll
  for (int q = 0; q < D / 16; ++q) {
    const uint4 m4 = reinterpret_cast<const uint4*>(mask + r * D)[q];
    mw[4 * q] = m4.x; mw[4 * q + 1] = m4.y; mw[4 * q + 2] = m4.z; mw[4 * q + 3] = m4.w;
  }
#pragma unroll
  for (int k = 0; k < D; ++k) {
    kp[k] = ((mw[k / 4] >> (8 * (k % 4))) & 0xffu) != 0;
    const float zz = __fadd_rn(lrelu(t1[k]), lrelu(t2[k]));
    z[k] = kp[k] ? zz / keep : 0.f;
    ss = fmaf(z[k], z[k], ss);
  }
  const float inv = 1.0f / sqrtf(fmaxf(ss, kNormEps));
  load_row<D>(d_out, r, ldo, g);                 // dLoss/d out_k row
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) dot = fmaf(g[k], __fmul_rn(z[k], inv), dot);
  float dz[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    float v = ss > kNormEps ? (g[k] - (z[k] * inv) * dot) * inv : g[k] * inv;
    if (d_ego_next) v += d_ego_next[r * D + k];
    dz[k] = kp[k] ? v / keep : 0.f;
  }
  float g1[D], g2[D], y1[D], y2[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    g1[k] = t1[k] > 0.f ? dz[k] : dz[k] * kLeaky;
    g2[k] = t2[k] > 0.f ? dz[k] : dz[k] * kLeaky;
  }
  RowMath<D>::affine_t(g1, s_w, y1);              // dT1·W_gcᵀ
  RowMath<D>::affine_t(g2, s_w + D * D, y2);      // dBi = dT2·W_biᵀ
  float ds[D], de[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    ds[k] = y1[k] + y2[k] * e[k];
    de[k] = y2[k] * s[k];
  }
  store_row<D>(dS, r, D, ds);
  store_row<D>(d_ego_direct, r, D, de);
  store_row<D>(dT1, r, D, g1);
  store_row<D>(dT2, r, D, g2);
}

// Weight gradients of one layer on the fp32 matrix cores (D == 16):
//   dW_gc = Sᵀ·dT1, dW_bi = (E⊙S)ᵀ·dT2, db_gc = Σ_rows dT1, db_bi = Σ_rows dT2.
// One wave per 256-row slab; v_mfma_f32_16x16x4_f32: lane l feeds A[i=l&15][k=l>>4] =
// X[row0+k][i] and B[k=l>>4][j=l&15] = G[row0+k][j] — four consecutive 64-byte rows per load.
__global__ __launch_bounds__(64) void ngcf_wgrad16_kernel(
    const float* __restrict__ ego, const float* __restrict__ S, const float* __restrict__ dT1,
    const float* __restrict__ dT2, int64_t n_rows, float* __restrict__ partial) {
  const int lane = threadIdx.x, c = lane & 15, kk = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * 256;
  f32x4 acc_g = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
  float sum1 = 0.f, sum2 = 0.f;
  for (int q = 0; q < 64; ++q) {
    const int64_t r = r0 + q * 4 + kk;
    float s = 0.f, e = 0.f, g1 = 0.f, g2 = 0.f;
    if (r < n_rows) {
      s = S[r * 16 + c]; e = ego[r * 16 + c]; g1 = dT1[r * 16 + c]; g2 = dT2[r * 16 + c];
    }
    acc_g = __builtin_amdgcn_mfma_f32_16x16x4f32(s, g1, acc_g, 0, 0, 0);
    acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(__fmul_rn(e, s), g2, acc_b, 0, 0, 0);
    sum1 += g1;
    sum2 += g2;
  }
  sum1 += __shfl_xor(sum1, 16, 64); sum1 += __shfl_xor(sum1, 32, 64);
  sum2 += __shfl_xor(sum2, 16, 64); sum2 += __shfl_xor(sum2, 32, 64);
  // C/D map of 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
  float* p = partial + (int64_t)blockIdx.x * 544;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    p[(kk * 4 + reg) * 16 + c] = acc_g[reg];
    p[256 + (kk * 4 + reg) * 16 + c] = acc_b[reg];
  }
  if (kk == 0) { p[512 + c] = sum1; p[528 + c] = sum2; }
}
// one wave per output element: lanes stride the slabs, fp64 accumulation, shuffle reduce
__global__ __launch_bounds__(256) void ngcf_wgrad_reduce_kernel(const float* __restrict__ partial,
                                                                int n_slabs, float* __restrict__ dWg,
                                                                float* __restrict__ dWb,
                                                                float* __restrict__ dbg,
                                                                float* __restrict__ dbb) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= 544) return;
  double acc = 0.0;
  for (int s = lane; s < n_slabs; s += 64) acc += (double)partial[(int64_t)s * 544 + i];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if (lane != 0) return;
  const float v = (float)acc;
  if (i < 256) dWg[i] = v;
  else if (i < 512) dWb[i - 256] = v;
  else if (i < 528) dbg[i - 512] = v;
  else dbb[i - 528] = v;
}

}  // namespace

extern "C" {

int nrhip_ngcf_layer_fwd(const float* d_ego, const float* d_S, const float* d_Wg,
                         const float* d_bg, const float* d_Wb, const float* d_bb, int64_t n_rows,
                         int d, float keep, uint8_t* d_mask, int mask_given, uint64_t seed,
                         uint64_t step, int layer, float* d_ego_out, float* d_out, int64_t ldo,
                         void* stream) {
  NR_REQUIRE(d_ego && d_S && d_Wg && d_bg && d_Wb && d_bb && d_mask && d_ego_out && d_out &&
                 n_rows >= 0 && ldo >= d && keep > 0.f && keep <= 1.f,
             NR_ERR_ARG, "ngcf_layer_fwd: bad arguments");
  NR_REQUIRE(d == 16, NR_ERR_UNSUPPORTED, "ngcf_layer: layer width %d not built (16)", d);
  NR_REQUIRE(ldo % 4 == 0 && ((uintptr_t)d_out % 16) == 0, NR_ERR_ARG,
             "ngcf_layer_fwd: output block must be 16-byte aligned with ldo %% 4 == 0");
  if (n_rows == 0) return NR_OK;
  hipLaunchKernelGGL(ngcf_layer_fwd_kernel<16>, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, d_ego, d_S, d_Wg, d_bg, d_Wb, d_bb, n_rows, keep, d_mask,
                     mask_given, seed, step, layer, d_ego_out, d_out, ldo);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_ngcf_layer_bwd(const float* d_ego, const float* d_S, const float* d_Wg,
                         const float* d_bg, const float* d_Wb, const float* d_bb, int64_t n_rows,
                         int d, float keep, const uint8_t* d_mask, const float* d_dout, int64_t ldo,
                         const float* d_dego_next, float* d_dS, float* d_dego_direct, float* d_dT1,
                         float* d_dT2, float* d_dWg, float* d_dbg, float* d_dWb, float* d_dbb,
                         void* d_ws, size_t ws_bytes, void* stream) {
  NR_REQUIRE(d_ego && d_S && d_Wg && d_bg && d_Wb && d_bb && d_mask && d_dout && d_dS &&
                 d_dego_direct && d_dT1 && d_dT2 && d_dWg && d_dbg && d_dWb && d_dbb && d_ws &&
                 n_rows >= 0 && ldo >= d,
             NR_ERR_ARG, "ngcf_layer_bwd: bad arguments");
  NR_REQUIRE(d == 16, NR_ERR_UNSUPPORTED, "ngcf_layer: layer width %d not built (16)", d);
  NR_REQUIRE(ldo % 4 == 0 && ((uintptr_t)d_dout % 16) == 0, NR_ERR_ARG,
             "ngcf_layer_bwd: gradient block must be 16-byte aligned with ldo %% 4 == 0");
  const int n_slabs = (int)((n_rows + 255) / 256);
  NR_REQUIRE(ws_bytes >= (size_t)(n_slabs > 0 ? n_slabs : 1) * 544 * sizeof(float), NR_ERR_WORKSPACE,
             "ngcf_layer_bwd: workspace too small");
  if (n_rows == 0) return NR_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ngcf_layer_bwd_kernel<16>, dim3((unsigned)n_slabs), dim3(256), 0, st, d_ego,
                     d_S, d_Wg, d_bg, d_Wb, d_bb, n_rows, keep, d_mask, d_dout, ldo, d_dego_next,
                     d_dS, d_dego_direct, d_dT1, d_dT2);
  NR_LAUNCH_CHECK();
  hipLaunchKernelGGL(ngcf_wgrad16_kernel, dim3((unsigned)n_slabs), dim3(64), 0, st, d_ego, d_S,
                     d_dT1, d_dT2, n_rows, (float*)d_ws);
  NR_LAUNCH_CHECK();
  hipLaunchKernelGGL(ngcf_wgrad_reduce_kernel, dim3(136), dim3(256), 0, st, (const float*)d_ws,
                     n_slabs, d_dWg, d_dWb, d_dbg, d_dbb);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_ngcf_workspace_bytes(int64_t n_rows, size_t* bytes) {
  NR_REQUIRE(bytes && n_rows >= 0, NR_ERR_ARG, "ngcf_workspace_bytes: bad arguments");
  *bytes = (size_t)((n_rows + 255) / 256 + 1) * 544 * sizeof(float);
  return NR_OK;
}

}  // extern "C"
