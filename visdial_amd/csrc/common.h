// Shared helpers for the visdial HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define VD_OK 0
#define VD_ERR_ARG -1
#define VD_ERR_HIP -2
#define VD_ERR_STATE -3

// thread-local message returned by vd_last_error()
void vd_set_error(const char* fmt, ...);

#define VD_CHECK_ARG(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) {                                                \
      vd_set_error(__VA_ARGS__);                                  \
      return VD_ERR_ARG;                                          \
    }                                                             \
  } while (0)

#define VD_HIP(expr)                                                          \
  do {                                                                        \
    hipError_t e__ = (expr);                                                  \
    if (e__ != hipSuccess) {                                                  \
      vd_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,              \
                   hipGetErrorString(e__));                                   \
      return VD_ERR_HIP;                                                      \
    }                                                                         \
  } while (0)

#define VD_LAUNCH_CHECK() VD_HIP(hipGetLastError())

static inline int vd_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float vd_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// wave64 all-reduce helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
