// common.h — shared helpers for the gfx950 kernels (internal; not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <mutex>

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

// Runs `f` once per DEVICE (kernel attributes such as the dynamic-LDS limit belong to the current device's copy of
// the code object): safe for a process that drives several GPUs from several threads.
struct OncePerDevice {
  std::mutex mu;
  unsigned long long done = 0;
  template <class F>
  void run(F &&f) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    std::lock_guard<std::mutex> lock(mu);
    if (!(done & bit)) {
      f();
      done |= bit;
    }
  }
};

// Which kernel families the calls of this process took (a3vt_dbg_path_counts; tests assert that a fixture REACHES the kernels it
// claims to pin).  Host-side counters, one relaxed increment per launch decision.
enum PathCount {
  PATH_STACK_QUAD = 0,   // a3vt_gcn_stack_fwd calls whose hidden layers ran the channel-sliced aggregation on hybrid rows
  PATH_STACK_ROWS,       // ... the half-wave row-walk aggregation on row-major rows
  PATH_RG_ADIRECT,       // rowgemm_kernel<19, ..., ADIRECT> launches (exact fp32, A operand straight into registers)
  PATH_RG3,              // rowgemm3_kernel launches (gemm mode 3, split operands)
  PATH_DW3,              // dw3_kernel launches
  PATH_DW_HYBRID,        // dw_kernel launches on quad-major operands
  PATH_RG16,             // rowgemm16_kernel launches (bf16 storage, weights in registers)
  PATH_STACK16_QUAD,     // bf16-storage stack forward calls on the channel-sliced aggregation
  PATH_RGW,              // rowgemmw_kernel launches (exact fp32, weights resident in registers; round 6)
  PATH_DWW,              // dww_kernel launches (exact fp32 dW on specialised waves; round 6)
  PATH_STACK_SPLIT,      // stack forward calls whose hidden layers aggregated with the P + bipartite split (gcn_csrqs.hip / csr16 split)
  PATH_CSR16T,           // csr16t_fwd_kernel launches (bf16 storage: tiled aggregation, neighbour rows from LDS; round 6)
  PATH_COUNT
};
void path_count(int which);

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

// 64-lane butterfly max; every lane gets the result.
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// ---- order-independent float accumulation (deterministic scatter-adds) ---------------------------------------------
// A float atomicAdd makes a sum depend on the order in which the threads arrive.  The backward scatters of this library
// (Chamfer y -> x term, surface-sample -> vertex, pooled-feature -> map) instead accumulate in 64-bit fixed point:
// integer addition is associative, so every order gives the same bits.  With |term| < 2^e for every term and at most
// 2^c terms per accumulator, terms are scaled by 2^(62 - e - c): the sum cannot overflow and its resolution is
// 2^-(62-c) of the largest term — about 2^-44 at 2^18 terms, far below the fp32 rounding of the terms themselves.
struct FixScale {
  int shift;  // term * 2^shift -> integer
};
// bound: any finite upper bound on |term| (> 0 or 0); count: upper bound on the number of terms per accumulator
__device__ __forceinline__ FixScale fix_scale(float bound, long long count) {
  int e = 0;
  (void)frexpf(bound, &e);  // bound = m * 2^e, 0.5 <= m < 1  ->  |term| <= bound < 2^e  (e = 0 for bound == 0)
  int c = 1;
  while ((1ll << c) <= count) ++c;
  return FixScale{62 - e - c};
}
__device__ __forceinline__ long long fix_from(float t, FixScale s) { return __float2ll_rn(ldexpf(t, s.shift)); }
__device__ __forceinline__ float fix_to(long long a, FixScale s) { return ldexpf((float)a, -s.shift); }
__device__ __forceinline__ void fix_add(long long *acc, long long v) {
  atomicAdd(reinterpret_cast<unsigned long long *>(acc), (unsigned long long)v);
}

}  // namespace a3vt
