// nr_common.h — shared plumbing for the HIP translation units of libneurec_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "nr_core.h"

#define NR_WAVE 64

// status codes returned across the C ABI (include/neurec_hip.h)
enum {
  NR_OK = 0,
  NR_ERR_ARG = 1,        // bad argument (maps to ValueError on the Python side)
  NR_ERR_UNSUPPORTED = 2, // shape/dtype outside what the kernels were built for
  NR_ERR_HIP = 3,        // HIP runtime failure
  NR_ERR_WORKSPACE = 4,  // workspace too small
};

extern "C" void nrhip_set_error(const char* fmt, ...);

#define NR_CHECK_HIP(expr)                                                     \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      nrhip_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,            \
                      hipGetErrorString(_e));                                  \
      return NR_ERR_HIP;                                                       \
    }                                                                          \
  } while (0)

#define NR_TRY(call)            \
  do {                          \
    const int rc_ = (call);     \
    if (rc_ != NR_OK) return rc_; \
  } while (0)
#define NR_REQUIRE(cond, code, ...)                                            \
  do {                                                                         \
    if (!(cond)) {                                                             \
      nrhip_set_error(__VA_ARGS__);                                            \
      return (code);                                                           \
    }                                                                          \
  } while (0)

#define NR_LAUNCH_CHECK()                                                      \
  do {                                                                         \
    hipError_t _e = hipGetLastError();                                         \
    if (_e != hipSuccess) {                                                    \
      nrhip_set_error("%s:%d: launch failed -> %s", __FILE__, __LINE__,        \
                      hipGetErrorString(_e));                                  \
      return NR_ERR_HIP;                                                       \
    }                                                                          \
  } while (0)

static inline size_t nr_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- wave-level helpers (wave64) -------------------------------------------
__device__ __forceinline__ int nr_lane() { return threadIdx.x & (NR_WAVE - 1); }

// Upward-rounded fp32 sum / product for error BOUNDS (gfx950 has no per-instruction rounding mode in HIP): the
// round-to-nearest result moved one ulp towards +inf is >= the exact value.  NaN / +inf pass through.
__device__ __forceinline__ float nr_next_up(float s) {
  if (!(s < __builtin_huge_valf())) return s;                       // NaN, +inf
  if (s == 0.f) return __uint_as_float(1u);                         // smallest sub-normal
  const uint32_t b = __float_as_uint(s);
  return __uint_as_float(s > 0.f ? b + 1u : b - 1u);
}
__device__ __forceinline__ float nr_add_up(float a, float b) { return nr_next_up(__fadd_rn(a, b)); }
__device__ __forceinline__ float nr_mul_up(float a, float b) { return nr_next_up(__fmul_rn(a, b)); }

__device__ __forceinline__ uint64_t nr_shfl_xor_u64(uint64_t x, int m) {
  uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  lo = __shfl_xor(lo, m, NR_WAVE);
  hi = __shfl_xor(hi, m, NR_WAVE);
  return ((uint64_t)hi << 32) | lo;
}
// all lanes receive the maximum
__device__ __forceinline__ uint64_t nr_wave_max_u64(uint64_t x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    uint64_t y = nr_shfl_xor_u64(x, m);
    x = y > x ? y : x;
  }
  return x;
}
__device__ __forceinline__ float nr_wave_sum_f32(float x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, NR_WAVE);
  return x;
}
__device__ __forceinline__ double nr_wave_sum_f64(double x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, NR_WAVE);
  return x;
}
// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int nr_mbcnt(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// Replay of TF-1.12 sparse Adam's zero-gradient steps from..upto on one table row held in registers
// (lane = column): the literal per-step sequence (nr::adam_sparse_tf with g = 0) with that step's
// lr_s.  The step sizes of up to 64 steps come in with ONE coalesced load (lane j holds step
// from + j) and are broadcast with v_readlane — a scalar load per iteration would put a memory
// round trip on every step of the chain.  Used by the lazy optimiser (adam.hip) and by the gradient
// kernel that must see rows as of the previous step (bpr.hip).
template <int CPL>
__device__ __forceinline__ void nr_lazy_replay(float (&w)[CPL], float (&mm)[CPL], float (&vv)[CPL],
                                               int from, int upto, const float* __restrict__ alpha_tab,
                                               int lane, float b1, float b2, float omb1, float omb2,
                                               float eps) {
  for (int s0 = from; s0 <= upto; s0 += NR_WAVE) {
    const int n = min(NR_WAVE, upto - s0 + 1);
    const float mine = alpha_tab[s0 + (lane < n ? lane : 0)];
    for (int j = 0; j < n; ++j) {
      const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), j));
#pragma unroll
      for (int c = 0; c < CPL; ++c) nr::adam_sparse_tf(0.f, w[c], mm[c], vv[c], a, b1, b2, omb1, omb2, eps);
    }
  }
}
