// chamfer.hip — bidirectional squared-L2 nearest neighbour + Chamfer reduction and gradient (gfx950).
//
// Replaces pytorch3d.loss.chamfer_distance(pred, gt, batch_reduction=None) as called at
// utility/utils.py:207,212 (knn_points K=1, squared distances, point_reduction="mean") and the
// mean over the 3 draws at utils.py:214-215.
//
// The search is brute force: 10k x 10k pairs per cloud pair cost 6.5 VALU lane-ops per pair here
// (3 sub, 1 mul, 2 fma, 1/2 min3; scalar fp32 — packed v_pk_*_f32 issues at half the rate on gfx950), the
// candidate cloud is broadcast from LDS, and each lane keeps R query points in registers.  The
// arg-min is recovered without carrying an index through the hot loop: the loop only tracks the
// best distance per 64-candidate chunk and remembers the winning chunk; that chunk is rescanned once
// at the end.  Inputs are tiny (120 KB per cloud): the kernel is VALU-bound, not HBM-bound.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace a3vt {

using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int kNNTile = 2048;  // candidates staged in LDS per pass (SoA, 24 KiB)
constexpr int kNNChunk = 64;   // arg-min granularity of the hot loop
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr float kFar = 1.0e18f;

__device__ __forceinline__ f32x2 sqdist2(float qx, float qy, float qz, f32x2 cx, f32x2 cy, f32x2 cz) {
  const f32x2 dx = f32x2{qx, qx} - cx, dy = f32x2{qy, qy} - cy, dz = f32x2{qz, qz} - cz;
  f32x2 d = dx * dx;
  d = __builtin_elementwise_fma(dy, dy, d);
  d = __builtin_elementwise_fma(dz, dz, d);
  return d;
}

// grid = (query blocks, clouds).  Cloud z: queries = q + (z % q_mod) * nq*3, candidates = c + (z % c_mod) * nc*3,
// outputs at z * nq.
template <int R>
__global__ __launch_bounds__(256) void nn_kernel(const float *__restrict__ q, int nq, int q_mod,
                                                 const float *__restrict__ c, int nc, int c_mod,
                                                 float *__restrict__ dist, int32_t *__restrict__ idx) {
  __shared__ __attribute__((aligned(16))) float sx[kNNTile], sy[kNNTile], sz[kNNTile];
  const int z = blockIdx.y;
  const float *qb = q + (long long)(z % q_mod) * nq * 3;
  const float *cb = c + (long long)(z % c_mod) * nc * 3;
  const int q0 = blockIdx.x * (256 * R) + threadIdx.x;

  float qx[R], qy[R], qz[R], best[R];
  int bchunk[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int qi = min(q0 + r * 256, nq - 1);
    qx[r] = qb[qi * 3 + 0];
    qy[r] = qb[qi * 3 + 1];
    qz[r] = qb[qi * 3 + 2];
    best[r] = 3.0e38f;
    bchunk[r] = 0;
  }

  for (int t0 = 0; t0 < nc; t0 += kNNTile) {
    __syncthreads();
    for (int i = threadIdx.x; i < kNNTile; i += 256) {
      const int ci = t0 + i;
      const bool ok = ci < nc;
      sx[i] = ok ? cb[ci * 3 + 0] : kFar;
      sy[i] = ok ? cb[ci * 3 + 1] : kFar;
      sz[i] = ok ? cb[ci * 3 + 2] : kFar;
    }
    __syncthreads();
    const int ntile = min(kNNTile, nc - t0);
    const int nchunk = (ntile + kNNChunk - 1) / kNNChunk;
    for (int ch = 0; ch < nchunk; ++ch) {
      float m[R];
#pragma unroll
      for (int r = 0; r < R; ++r) m[r] = 3.0e38f;
      const int j0 = ch * kNNChunk;
#pragma unroll 4
      for (int j = 0; j < kNNChunk; j += 4) {
        const f32x4 cx = *reinterpret_cast<const f32x4 *>(sx + j0 + j);
        const f32x4 cy = *reinterpret_cast<const f32x4 *>(sy + j0 + j);
        const f32x4 cz = *reinterpret_cast<const f32x4 *>(sz + j0 + j);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          // scalar fp32 ops: v_pk_*_f32 issues at half the rate on gfx950 (tools/ubench/valu_peak.hip) and drags s_nop
          // hazards along; same fused arithmetic as sqdist2, two v_min3 per four candidates
          float d[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float dx = qx[r] - cx[u], dy = qy[r] - cy[u], dz = qz[r] - cz[u];
            d[u] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
          }
          m[r] = __builtin_fminf(__builtin_fminf(m[r], d[0]), d[1]);
          m[r] = __builtin_fminf(__builtin_fminf(m[r], d[2]), d[3]);
        }
      }
      const int gch = (t0 + j0) / kNNChunk;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (m[r] < best[r]) {  // strict: the earliest chunk holding the minimum wins
          best[r] = m[r];
          bchunk[r] = gch;
        }
      }
    }
  }

  // Rescan the winning chunk (same arithmetic) for the first arg-min.
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int qi = q0 + r * 256;
    if (qi >= nq) continue;
    const int j0 = bchunk[r] * kNNChunk;
    float bd = 3.0e38f;
    int bj = j0;
    for (int j = 0; j < kNNChunk; j += 2) {
      const int ja = j0 + j, jb = j0 + j + 1;
      const f32x2 cx = {ja < nc ? cb[ja * 3 + 0] : kFar, jb < nc ? cb[jb * 3 + 0] : kFar};
      const f32x2 cy = {ja < nc ? cb[ja * 3 + 1] : kFar, jb < nc ? cb[jb * 3 + 1] : kFar};
      const f32x2 cz = {ja < nc ? cb[ja * 3 + 2] : kFar, jb < nc ? cb[jb * 3 + 2] : kFar};
      const f32x2 d = sqdist2(qx[r], qy[r], qz[r], cx, cy, cz);
      if (d[0] < bd) { bd = d[0]; bj = ja; }
      if (d[1] < bd) { bd = d[1]; bj = jb; }
    }
    dist[(long long)z * nq + qi] = bd;
    idx[(long long)z * nq + qi] = bj;
  }
}

template <int R>
static void launch_nn_r(const float *q, int nq, int q_mod, const float *c, int nc, int c_mod, int nclouds, float *dist,
                        int32_t *idx, hipStream_t s) {
  dim3 grid(cdiv(nq, 256 * R), nclouds);
  A3VT_LAUNCH((nn_kernel<R>), grid, dim3(256), 0, s, q, nq, q_mod, c, nc, c_mod, dist, idx);
}

static int launch_nn(const float *q, int nq, int q_mod, const float *c, int nc, int c_mod, int nclouds, float *dist,
                     int32_t *idx, hipStream_t s) {
  // R queries per lane.  The kernel is VALU bound, so what sets its time is the busiest CU: workgroups per CU (rounded
  // up) x R; among equals the smaller R wins (more waves per SIMD: R = 10 measured 5.5 ms where R = 5 takes 4.85, R = 8 5.2).
  // 10,000-point clouds x 192: R = 8 gives 960 workgroups = 3.75 per CU, so some CUs carry 4 x 8 = 32 units; R = 5
  // gives 1,536 = exactly 6 per CU, 30 units, all resident at once (72 VGPRs) — measured 5.2 -> 4.85 ms per call.
  static const int env_r = getenv("A3VT_NN_R") ? atoi(getenv("A3VT_NN_R")) : 0;  // developer override
  int best_r = 4;
  long long best_cost = -1;
  for (int r : {5, 4, 6, 3}) {  // 8 and 10 can never beat 4 and 5 under this cost (half the R, at most twice the workgroups)
    const long long wgs = (long long)cdiv(nq, 256 * r) * nclouds;
    const long long cost = ((wgs + 255) / 256) * r;
    if (best_cost < 0 || cost < best_cost) best_cost = cost, best_r = r;
  }
  if (env_r) best_r = env_r;
  switch (best_r) {
    case 6: launch_nn_r<6>(q, nq, q_mod, c, nc, c_mod, nclouds, dist, idx, s); break;
    case 5: launch_nn_r<5>(q, nq, q_mod, c, nc, c_mod, nclouds, dist, idx, s); break;
    case 3: launch_nn_r<3>(q, nq, q_mod, c, nc, c_mod, nclouds, dist, idx, s); break;
    default: launch_nn_r<4>(q, nq, q_mod, c, nc, c_mod, nclouds, dist, idx, s); break;
  }
  A3VT_CHECK_LAUNCH();
  return 0;
}

// cd[b] = (1/draws) * sum_r ( mean_i dxy[r][b][i] + mean_j dyx[r][b][j] ) ; one workgroup per b, fixed order.
__global__ __launch_bounds__(256) void chamfer_reduce_kernel(const float *__restrict__ dxy,
                                                             const float *__restrict__ dyx, int draws, int batch,
                                                             int p, int q, float *__restrict__ cd) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float total = 0.f;
  for (int r = 0; r < draws; ++r) {
    float s1 = 0.f, s2 = 0.f;
    const float *a = dxy + ((long long)r * batch + b) * p;
    const float *bb = dyx + ((long long)r * batch + b) * q;
    for (int i = threadIdx.x; i < p; i += 256) s1 += a[i];
    for (int i = threadIdx.x; i < q; i += 256) s2 += bb[i];
    float v = s1 / (float)p + s2 / (float)q;
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    total += (red[0] + red[1]) + (red[2] + red[3]);
  }
  if (threadIdx.x == 0) cd[b] = total / (float)draws;
}

int launch_chamfer_fwd(const float *x, const float *y, int draws, int batch, int p, int q, float *dxy, int32_t *ixy,
                       float *dyx, int32_t *iyx, float *cd, hipStream_t s) {
  if (p <= 0 || q <= 0 || draws <= 0 || batch <= 0) {
    set_error("chamfer_fwd: empty input (draws=%d batch=%d p=%d q=%d)", draws, batch, p, q);
    return -1;
  }
  // x -> y: queries = x clouds (draws*batch distinct), candidates = y[b]
  if (int rc = launch_nn(x, p, draws * batch, y, q, batch, draws * batch, dxy, ixy, s)) return rc;
  // y -> x: queries = y[b], candidates = x[r][b]
  if (int rc = launch_nn(y, q, batch, x, p, draws * batch, draws * batch, dyx, iyx, s)) return rc;
  A3VT_LAUNCH(chamfer_reduce_kernel, dim3(batch), dim3(256), 0, s, dxy, dyx, draws, batch, p, q, cd);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// Gradient.  With g = grad_cd[b] / draws:
//   gx[r][b][i]  = g * (2/P) (x_i - y_nn(i))                       (direct store, kernel A)
//   gx[r][b][nn(j)] += g * (2/Q) (x_nn(j) - y_j)                   (atomic, kernel B)
//   gy[b][nn(i)] -= first term ;  gy[b][j] -= second term          (atomic, optional)
__global__ __launch_bounds__(256) void chamfer_bwd_x_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                            int draws, int batch, int p, int q,
                                                            const int32_t *__restrict__ ixy,
                                                            const float *__restrict__ gcd, float *__restrict__ gx,
                                                            float *__restrict__ gy) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (r, b, i)
  const long long total = (long long)draws * batch * p;
  if (i >= total) return;
  const int b = (int)((i / p) % batch);
  const float coef = gcd[b] / (float)draws * (2.0f / (float)p);
  const long long yj = (long long)b * q + ixy[i];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float t = coef * (x[i * 3 + d] - y[yj * 3 + d]);
    gx[i * 3 + d] = t;
    if (gy) atomicAdd(gy + yj * 3 + d, -t);
  }
}

__global__ __launch_bounds__(256) void chamfer_bwd_y_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                            int draws, int batch, int p, int q,
                                                            const int32_t *__restrict__ iyx,
                                                            const float *__restrict__ gcd, float *__restrict__ gx,
                                                            float *__restrict__ gy) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (r, b, j)
  const long long total = (long long)draws * batch * q;
  if (j >= total) return;
  const long long rb = j / q;
  const int b = (int)(rb % batch);
  const int jj = (int)(j - rb * q);
  const float coef = gcd[b] / (float)draws * (2.0f / (float)q);
  const long long xi = rb * p + iyx[j];
  const long long yj = (long long)b * q + jj;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float t = coef * (x[xi * 3 + d] - y[yj * 3 + d]);
    atomicAdd(gx + xi * 3 + d, t);
    if (gy) atomicAdd(gy + yj * 3 + d, -t);
  }
}

int launch_chamfer_bwd(const float *x, const float *y, int draws, int batch, int p, int q, const int32_t *ixy,
                       const int32_t *iyx, const float *gcd, float *gx, float *gy, hipStream_t s) {
  if (gy)
    if (int rc = launch_fill_zero(gy, (size_t)batch * q * 3, s)) return rc;
  const long long tx = (long long)draws * batch * p, ty = (long long)draws * batch * q;
  A3VT_LAUNCH(chamfer_bwd_x_kernel, dim3(cdiv(tx, 256)), dim3(256), 0, s, x, y, draws, batch, p, q, ixy, gcd, gx,
                     gy);
  A3VT_CHECK_LAUNCH();
  A3VT_LAUNCH(chamfer_bwd_y_kernel, dim3(cdiv(ty, 256)), dim3(256), 0, s, x, y, draws, batch, p, q, iyx, gcd, gx,
                     gy);
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace a3vt
