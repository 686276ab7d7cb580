// Shared helpers for the gfx950 kernels of libhydra_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/hydra_hip.h"
#include "../../include/hydra_hip_diag.h"

namespace hydra {

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// thread-local error text returned by hydra_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

#define HYDRA_CHECK_ARG(cond, ...)                         \
  do {                                                     \
    if (!(cond)) return hydra::fail(HYDRA_ERR_ARG, __VA_ARGS__); \
  } while (0)

#define HYDRA_CHECK_LAUNCH(name)                                                         \
  do {                                                                                   \
    hipError_t e__ = hipGetLastError();                                                  \
    if (e__ != hipSuccess)                                                               \
      return hydra::fail(HYDRA_ERR_HIP, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// opt a kernel in to more than 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU)
static inline int reserve_lds(const void* kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return HYDRA_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess)
    return fail(HYDRA_ERR_HIP, "cannot reserve %zu bytes of LDS: %s", bytes, hipGetErrorString(e));
  return HYDRA_OK;
}

// complex helpers on (re, im) pairs kept as f64x2 so loads/stores are 16 bytes
__device__ __forceinline__ f64x2 cmul(f64x2 a, f64x2 b) {
  f64x2 r;
  r.x = a.x * b.x - a.y * b.y;
  r.y = a.x * b.y + a.y * b.x;
  return r;
}
// conj(a) * b
__device__ __forceinline__ f64x2 cmul_conj(f64x2 a, f64x2 b) {
  f64x2 r;
  r.x = a.x * b.x + a.y * b.y;
  r.y = a.x * b.y - a.y * b.x;
  return r;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

}  // namespace hydra
