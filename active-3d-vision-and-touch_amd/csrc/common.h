// common.h — shared helpers for the gfx950 kernels (internal; not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace a3vt {

void set_error(const char *fmt, ...);

#define A3VT_CHECK_ARG(cond)                                                              \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      a3vt::set_error("%s:%d: argument check failed: %s", __FILE__, __LINE__, #cond);     \
      return -1;                                                                          \
    }                                                                                     \
  } while (0)

// Launch after clearing any stale error left in the thread by earlier HIP calls of the host program
// (hipGetLastError is sticky per thread; torch's own probing can leave e.g. hipErrorNoDevice behind).
#define A3VT_LAUNCH(...)            \
  do {                              \
    (void)hipGetLastError();        \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

#define A3VT_CHECK_LAUNCH()                                                               \
  do {                                                                                    \
    hipError_t e__ = hipGetLastError();                                                   \
    if (e__ != hipSuccess) {                                                              \
      a3vt::set_error("%s:%d: HIP launch failed: %s", __FILE__, __LINE__,                 \
                      hipGetErrorString(e__));                                            \
      return -2;                                                                          \
    }                                                                                     \
  } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int pad4(int n) { return (n + 3) & ~3; }
static inline int pad16(int n) { return (n + 15) & ~15; }

constexpr int kWave = 64;  // gfx950 wavefront

// 64-lane butterfly sum; every lane gets the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace a3vt
