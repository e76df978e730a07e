// bias_grad.hip — the bias gradient of a convolution whose output gradient is stored channels-last:
//   db[c] = sum over (n, h, w) of dY[n, h, w, c]                 (dY: [rows = N*H*W][C], fp32 or bf16; db: fp32)
//
// Replaces, on the image branch of the bf16 configurations, the column reduction torch's convolution backward
// issues for it (`grad_output.sum((0, 2, 3))`: 1.28 ms for the 25 MB gradient of the 3-channel full-resolution map,
// 2.9 ms per step over the pyramid, profiles/r03_config3_bf16s_steady_kernels.txt).  Reference: the `nn.Conv2d`
// layers of `CNN_layer` / `Image_Encoder`, pterotactyl/reconstruction/vision/model.py:15-47 (their bias gradients).
//
// HBM-bound: every element is read once in 16-byte pieces.  A thread walks the pieces gtid, gtid + S, ... where the
// stride S (in elements, 8*S) is a multiple of C, so each of its eight accumulators stays on ONE channel for the
// whole walk; the workgroup folds its 2048 accumulators into C partial sums in a fixed order (no atomics: the
// result does not depend on scheduling), and the per-workgroup rows are summed by slab_reduce.
#include "common.h"
#include "kernels.h"

namespace a3vt {

namespace {

constexpr int kBgThreads = 256;
constexpr int kBgMaxWgs = 1024;

__device__ __forceinline__ float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

template <bool BF16>
__device__ __forceinline__ void load_piece(const void *g, long long piece, long long n_elem, float v[8]) {
  const long long e0 = piece * 8;
  if (e0 + 8 <= n_elem) {
    if (BF16) {
      const uint4 u = *reinterpret_cast<const uint4 *>(static_cast<const uint16_t *>(g) + e0);
      v[0] = bf16_lo(u.x); v[1] = bf16_hi(u.x); v[2] = bf16_lo(u.y); v[3] = bf16_hi(u.y);
      v[4] = bf16_lo(u.z); v[5] = bf16_hi(u.z); v[6] = bf16_lo(u.w); v[7] = bf16_hi(u.w);
    } else {
      const float4 a = *reinterpret_cast<const float4 *>(static_cast<const float *>(g) + e0);
      const float4 b = *reinterpret_cast<const float4 *>(static_cast<const float *>(g) + e0 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
  } else {  // the last, partial piece
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = 0.f;
      if (e0 + j < n_elem)
        x = BF16 ? bf16_lo(static_cast<const uint16_t *>(g)[e0 + j]) : static_cast<const float *>(g)[e0 + j];
      v[j] = x;
    }
  }
}

template <bool BF16>
__global__ __launch_bounds__(kBgThreads) void bias_grad_kernel(const void *__restrict__ g, long long n_elem, int c,
                                                               float *__restrict__ slab) {
  __shared__ float part[kBgThreads][9];
  const int t = threadIdx.x;
  const long long stride = (long long)gridDim.x * kBgThreads;   // pieces; 8 * stride is a multiple of c (host)
  const long long n_piece = (n_elem + 7) / 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  long long p = (long long)blockIdx.x * kBgThreads + t;
  for (; p + 3 * stride < n_piece; p += 4 * stride) {   // four pieces in flight
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) load_piece<BF16>(g, p + u * stride, n_elem, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[u][j];
  }
  for (; p < n_piece; p += stride) {
    float v[8];
    load_piece<BF16>(g, p, n_elem, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[t][j] = acc[j];
  __syncthreads();
  // accumulator j of thread u holds channel (base + 8u + j) mod c
  const int base = (int)(((long long)blockIdx.x * kBgThreads * 8) % c);
  for (int ch = t; ch < c; ch += kBgThreads) {
    int x = ch - base;
    x = x < 0 ? x + c : x;                       // the j of thread 0 that lands on this channel (mod c)
    float s = 0.f;
    for (int u = 0; u < kBgThreads; ++u) {
      for (int j = x; j < 8; j += c) s += part[u][j];
      x -= 8 % c;
      x = x < 0 ? x + c : x;
    }
    slab[(size_t)blockIdx.x * c + ch] = s;
  }
}

}  // namespace

// Workgroups: as many as the data feeds (one piece per thread at least), at most kBgMaxWgs, and a multiple of
// c / gcd(c, 2048) so that a thread's stride is whole channels' periods.
int bias_grad_wgs(long long n_elem, int c) {
  long long a = c, b = 2048;
  while (b) { const long long r = a % b; a = b; b = r; }
  const int m = (int)(c / a);
  if (m > kBgMaxWgs) return 0;
  const long long want = (n_elem / 8 + kBgThreads - 1) / kBgThreads;
  long long n = want < kBgMaxWgs ? want : kBgMaxWgs;
  n = n / m * m;
  return (int)(n < m ? m : n);
}

int launch_bias_grad(const void *g, int bf16, long long rows, int c, float *out, float *slab, hipStream_t s) {
  const long long n_elem = rows * c;
  const int nwg = bias_grad_wgs(n_elem, c);
  A3VT_CHECK_ARG(nwg > 0);
  if (bf16)
    A3VT_LAUNCH(bias_grad_kernel<true>, dim3(nwg), dim3(kBgThreads), 0, s, g, n_elem, c, slab);
  else
    A3VT_LAUNCH(bias_grad_kernel<false>, dim3(nwg), dim3(kBgThreads), 0, s, g, n_elem, c, slab);
  A3VT_CHECK_LAUNCH();
  return launch_slab_reduce(slab, nwg, (size_t)c, (size_t)c, out, s);
}

}  // namespace a3vt
