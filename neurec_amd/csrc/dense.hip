// dense.hip — the dense half of an NGCF propagation layer (second-model coverage, config 5).
//
// Stands in for the TF ops of NGCF._create_ngcf_embed (model/general_recommender/NGCF.py:181-198)
// and their autodiff.  Per layer, with S = Â·E from the SpMM kernel (NGCF.py:174-179):
//     T1 = S·W_gc + b_gc            T2 = (E ⊙ S)·W_bi + b_bi
//     Z  = leaky_relu(T1) + leaky_relu(T2)                  (alpha = 0.2)
//     E' = dropout(Z, keep) = Z / keep · mask               (always on, also at evaluation: NGCF.py:193)
//     out_k = l2_normalize(E', axis=1)                       (x · rsqrt(max(Σx², 1e-12)))
// The layer width is 16 (conf/NGCF.properties: embedding_size=16, layer_size=[16,16]); a row is
// 64 bytes: FOUR LANES own one node row (four output columns each), the two 16×16 weight
// matrices sit in LDS, and the row-wise normalisation is a quad shuffle.  Everything
// here is a streaming pass over [N][16] buffers (4.5 MB at gowalla): HBM/launch-bound, fused so
// that a layer is one forward kernel and (backward) one row kernel + one weight-gradient kernel.
//
// Weight gradients dW = Xᵀ·G (X, G: [N][16]) are tall-skinny products reduced over the N rows:
// they run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, one wave per 256-row slab) and are
// combined over slabs in slab order (deterministic).
#include "nr_common.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kLeaky = 0.2f;
constexpr float kNormEps = 1e-12f;

__device__ __forceinline__ float lrelu(float x) { return x > 0.f ? x : __fmul_rn(x, kLeaky); }

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
__device__ __forceinline__ void stage_weights(const float* Wg, const float* bg, const float* Wb,
                                              const float* bb, float* s) {
  for (int i = threadIdx.x; i < D * D; i += blockDim.x) { s[i] = Wg[i]; s[D * D + i] = Wb[i]; }
  for (int i = threadIdx.x; i < D; i += blockDim.x) { s[2 * D * D + i] = bg[i]; s[2 * D * D + D + i] = bb[i]; }
  __syncthreads();
}

// ---- the layer kernels: FOUR lanes per node row (D = 16) ------------------------------------------
// forward of one layer; mask_io: if mask_given it is read, else it is drawn and written.
// Lane q of a quad computes output columns 4q..4q+3: the row's inputs are read by all four lanes (one 64-byte line),
// a weight row is one 16-byte LDS read, whole-row quantities (the norm, the dot with the incoming gradient, the
// transposed products) take the other lanes' values by quad shuffles; every output is a k-ascending fmaf chain.
// (r01's one-thread-per-row form held seven 16-float rows and 512 hoisted weights per lane — 256 VGPRs + scratch,
//  one wave per SIMD, 29 / 37 us for 4.5 MB buffers; this form is bit-identical at a fifth of the time.  It left the
//  product in r06; profiles/r02_config5_ngcf_multivae.json has the A/B.)
constexpr int kQuadRows = 64;                      // rows per 256-thread workgroup

__device__ __forceinline__ void quad_gather(const float (&mine)[4], float (&all)[16], int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int c = 0; c < 4; ++c) all[4 * q + c] = __shfl(mine[c], (lane & ~3) | q, NR_WAVE);
}
// T[4q..4q+3] = x·W[:, 4q..4q+3] + b: k-ascending chain per column, one float4 of W per k
__device__ __forceinline__ void quad_affine(const float (&x)[16], const float* W, const float* b, int q,
                                            float (&t)[4]) {
  const float4 b4 = *reinterpret_cast<const float4*>(b + 4 * q);
  t[0] = b4.x; t[1] = b4.y; t[2] = b4.z; t[3] = b4.w;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float4 w = *reinterpret_cast<const float4*>(W + k * 16 + 4 * q);
    t[0] = fmaf(x[k], w.x, t[0]); t[1] = fmaf(x[k], w.y, t[1]);
    t[2] = fmaf(x[k], w.z, t[2]); t[3] = fmaf(x[k], w.w, t[3]);
  }
}
// y[4q..4q+3] = g·Wᵀ rows 4q..4q+3: j-ascending chain per output
__device__ __forceinline__ void quad_affine_t(const float (&g)[16], const float* W, int q, float (&y)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float* wr = W + (4 * q + c) * 16;
    float acc = 0.f;
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      const float4 w = *reinterpret_cast<const float4*>(wr + 4 * j4);
      acc = fmaf(g[4 * j4], w.x, acc); acc = fmaf(g[4 * j4 + 1], w.y, acc);
      acc = fmaf(g[4 * j4 + 2], w.z, acc); acc = fmaf(g[4 * j4 + 3], w.w, acc);
    }
    y[c] = acc;
  }
}
__device__ __forceinline__ void load_row16(const float* __restrict__ p, int64_t row, int64_t ld, float (&x)[16]) {
  load_row<16>(p, row, ld, x);
}

__global__ __launch_bounds__(4 * kQuadRows) void ngcf_layer_fwd_quad_kernel(
    const float* __restrict__ ego, const float* __restrict__ S, const float* __restrict__ Wg,
    const float* __restrict__ bg, const float* __restrict__ Wb, const float* __restrict__ bb,
    int64_t n_rows, float keep, uint8_t* __restrict__ mask_io, int mask_given, uint64_t seed,
    uint64_t step, int layer, float* __restrict__ ego_out, float* __restrict__ out, int64_t ldo) {
  constexpr int D = 16;
  __shared__ __attribute__((aligned(16))) float s_w[2 * D * D + 2 * D];
  stage_weights<D>(Wg, bg, Wb, bb, s_w);
  const int lane = threadIdx.x & 63, q = threadIdx.x & 3;
  const int64_t r = (int64_t)blockIdx.x * kQuadRows + (threadIdx.x >> 2);
  const bool live = r < n_rows;
  const int64_t rr = live ? r : 0;                      // idle quads compute on row 0 (shuffles stay uniform), store nothing
  float e[D], sv[D], bi[D], t1[4], t2[4];
  load_row16(ego, rr, D, e);
  load_row16(S, rr, D, sv);
#pragma unroll
  for (int k = 0; k < D; ++k) bi[k] = __fmul_rn(e[k], sv[k]);
  quad_affine(sv, s_w, s_w + 2 * D * D, q, t1);
  quad_affine(bi, s_w + D * D, s_w + 2 * D * D + D, q, t2);
  uint32_t mw;                                           // the mask bytes of columns 4q..4q+3
  if (mask_given) {
    mw = *reinterpret_cast<const uint32_t*>(mask_io + rr * D + 4 * q);
  } else {
    const uint64_t key = nr::splitmix64(seed ^ (step * 0x9e3779b97f4a7c15ull + layer));
    const uint64_t hsh = nr::splitmix64(key ^ ((uint64_t)rr * (D / 4) + q));
    mw = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if ((float)((hsh >> (16 * i)) & 0xffffu) * (1.0f / 65536.0f) < keep) mw |= 1u << (8 * i);
    if (live) *reinterpret_cast<uint32_t*>(mask_io + r * D + 4 * q) = mw;
  }
  float z[4], zall[D];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bool kp = ((mw >> (8 * c)) & 0xffu) != 0;
    const float zz = __fadd_rn(lrelu(t1[c]), lrelu(t2[c]));
    z[c] = kp ? zz / keep : 0.f;
  }
  quad_gather(z, zall, lane);
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) ss = fmaf(zall[k], zall[k], ss);
  const float inv = 1.0f / sqrtf(fmaxf(ss, kNormEps));
  if (!live) return;
  *reinterpret_cast<float4*>(ego_out + r * D + 4 * q) = make_float4(z[0], z[1], z[2], z[3]);
  *reinterpret_cast<float4*>(out + r * ldo + 4 * q) =
      make_float4(__fmul_rn(z[0], inv), __fmul_rn(z[1], inv), __fmul_rn(z[2], inv), __fmul_rn(z[3], inv));
}

__global__ __launch_bounds__(4 * kQuadRows) void ngcf_layer_bwd_quad_kernel(
    const float* __restrict__ ego, const float* __restrict__ S, const float* __restrict__ Wg,
    const float* __restrict__ bg, const float* __restrict__ Wb, const float* __restrict__ bb,
    int64_t n_rows, float keep, const uint8_t* __restrict__ mask, const float* __restrict__ d_out,
    int64_t ldo, const float* __restrict__ d_ego_next, float* __restrict__ dS,
    float* __restrict__ d_ego_direct, float* __restrict__ dT1, float* __restrict__ dT2) {
  constexpr int D = 16;
  __shared__ __attribute__((aligned(16))) float s_w[2 * D * D + 2 * D];
  stage_weights<D>(Wg, bg, Wb, bb, s_w);
  const int lane = threadIdx.x & 63, q = threadIdx.x & 3;
  const int64_t r = (int64_t)blockIdx.x * kQuadRows + (threadIdx.x >> 2);
  const bool live = r < n_rows;
  const int64_t rr = live ? r : 0;
  float e[D], sv[D], bi[D], t1[4], t2[4], g[D];
  load_row16(ego, rr, D, e);
  load_row16(S, rr, D, sv);
  load_row16(d_out, rr, ldo, g);                         // dLoss/d out_k row
#pragma unroll
  for (int k = 0; k < D; ++k) bi[k] = __fmul_rn(e[k], sv[k]);
  quad_affine(sv, s_w, s_w + 2 * D * D, q, t1);
  quad_affine(bi, s_w + D * D, s_w + 2 * D * D + D, q, t2);
  const uint32_t mw = *reinterpret_cast<const uint32_t*>(mask + rr * D + 4 * q);
  float z[4], zall[D];
  bool kp[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    kp[c] = ((mw >> (8 * c)) & 0xffu) != 0;
    const float zz = __fadd_rn(lrelu(t1[c]), lrelu(t2[c]));
    z[c] = kp[c] ? zz / keep : 0.f;
  }
  quad_gather(z, zall, lane);
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) ss = fmaf(zall[k], zall[k], ss);
  const float inv = 1.0f / sqrtf(fmaxf(ss, kNormEps));
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) dot = fmaf(g[k], __fmul_rn(zall[k], inv), dot);
  float g1[4], g2[4];
  float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
  if (d_ego_next) nxt = *reinterpret_cast<const float4*>(d_ego_next + rr * D + 4 * q);
  const float nx[4] = {nxt.x, nxt.y, nxt.z, nxt.w};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float gk = g[4 * q + c];
    float v = ss > kNormEps ? (gk - (z[c] * inv) * dot) * inv : gk * inv;
    if (d_ego_next) v += nx[c];
    const float dz = kp[c] ? v / keep : 0.f;
    g1[c] = t1[c] > 0.f ? dz : dz * kLeaky;
    g2[c] = t2[c] > 0.f ? dz : dz * kLeaky;
  }
  float g1all[D], g2all[D], y1[4], y2[4];
  quad_gather(g1, g1all, lane);
  quad_gather(g2, g2all, lane);
  quad_affine_t(g1all, s_w, q, y1);                       // dT1·W_gcᵀ
  quad_affine_t(g2all, s_w + D * D, q, y2);               // dBi = dT2·W_biᵀ
  if (!live) return;
  float ds[4], de[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    ds[c] = y1[c] + y2[c] * e[4 * q + c];
    de[c] = y2[c] * sv[4 * q + c];
  }
  *reinterpret_cast<float4*>(dS + r * D + 4 * q) = make_float4(ds[0], ds[1], ds[2], ds[3]);
  *reinterpret_cast<float4*>(d_ego_direct + r * D + 4 * q) = make_float4(de[0], de[1], de[2], de[3]);
  *reinterpret_cast<float4*>(dT1 + r * D + 4 * q) = make_float4(g1[0], g1[1], g1[2], g1[3]);
  *reinterpret_cast<float4*>(dT2 + r * D + 4 * q) = make_float4(g2[0], g2[1], g2[2], g2[3]);
}

// Weight gradients of one layer on the fp32 matrix cores (D == 16):
//   dW_gc = Sᵀ·dT1, dW_bi = (E⊙S)ᵀ·dT2, db_gc = Σ_rows dT1, db_bi = Σ_rows dT2.
// One wave per kWgradSlab-row slab; v_mfma_f32_16x16x4_f32: lane l feeds A[i=l&15][k=l>>4] =
// X[row0+k][i] and B[k=l>>4][j=l&15] = G[row0+k][j] — four consecutive 64-byte rows per load.
// A slab is 64 rows and ALL its 64 loads per lane are requested before the first MFMA: the first
// version walked 256 rows per wave, four loads and two MFMAs per dependent round trip (277 waves,
// 23 us for 18 MB).
constexpr int kWgradSlab = 64;
__global__ __launch_bounds__(64) void ngcf_wgrad16_kernel(
    const float* __restrict__ ego, const float* __restrict__ S, const float* __restrict__ dT1,
    const float* __restrict__ dT2, int64_t n_rows, float* __restrict__ partial) {
  const int lane = threadIdx.x, c = lane & 15, kk = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * kWgradSlab;
  constexpr int Q = kWgradSlab / 4;
  float s[Q], e[Q], g1[Q], g2[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int64_t r = r0 + q * 4 + kk;
    const int64_t rr = r < n_rows ? r : 0;
    const float live = r < n_rows ? 1.f : 0.f;
    s[q] = S[rr * 16 + c] * live; e[q] = ego[rr * 16 + c];
    g1[q] = dT1[rr * 16 + c] * live; g2[q] = dT2[rr * 16 + c] * live;
  }
  f32x4 acc_g = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
  float sum1 = 0.f, sum2 = 0.f;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    acc_g = __builtin_amdgcn_mfma_f32_16x16x4f32(s[q], g1[q], acc_g, 0, 0, 0);
    acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(__fmul_rn(e[q], s[q]), g2[q], acc_b, 0, 0, 0);
    sum1 += g1[q];
    sum2 += g2[q];
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
  hipLaunchKernelGGL(ngcf_layer_fwd_quad_kernel, dim3((unsigned)((n_rows + kQuadRows - 1) / kQuadRows)),
                       dim3(4 * kQuadRows), 0, (hipStream_t)stream, d_ego, d_S, d_Wg, d_bg, d_Wb, d_bb, n_rows,
                       keep, d_mask, mask_given, seed, step, layer, d_ego_out, d_out, ldo);
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
  const int w_slabs = (int)((n_rows + kWgradSlab - 1) / kWgradSlab);
  NR_REQUIRE(ws_bytes >= (size_t)(w_slabs > 0 ? w_slabs : 1) * 544 * sizeof(float), NR_ERR_WORKSPACE,
             "ngcf_layer_bwd: workspace too small");
  if (n_rows == 0) return NR_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ngcf_layer_bwd_quad_kernel, dim3((unsigned)((n_rows + kQuadRows - 1) / kQuadRows)),
                       dim3(4 * kQuadRows), 0, st, d_ego, d_S, d_Wg, d_bg, d_Wb, d_bb, n_rows, keep, d_mask,
                       d_dout, ldo, d_dego_next, d_dS, d_dego_direct, d_dT1, d_dT2);
  NR_LAUNCH_CHECK();
  hipLaunchKernelGGL(ngcf_wgrad16_kernel, dim3((unsigned)w_slabs), dim3(64), 0, st, d_ego, d_S,
                     d_dT1, d_dT2, n_rows, (float*)d_ws);
  NR_LAUNCH_CHECK();
  hipLaunchKernelGGL(ngcf_wgrad_reduce_kernel, dim3(136), dim3(256), 0, st, (const float*)d_ws,
                     w_slabs, d_dWg, d_dWb, d_dbg, d_dbb);
  NR_LAUNCH_CHECK();
  return NR_OK;
}

int nrhip_ngcf_workspace_bytes(int64_t n_rows, size_t* bytes) {
  NR_REQUIRE(bytes && n_rows >= 0, NR_ERR_ARG, "ngcf_workspace_bytes: bad arguments");
  *bytes = (size_t)((n_rows + kWgradSlab - 1) / kWgradSlab + 1) * 544 * sizeof(float);
  return NR_OK;
}

}  // extern "C"
