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
