// nn_prune.hip — EXACT squared-L2 nearest neighbour with spatial pruning (gfx950).
//
// Same results, bit for bit, as the brute-force searches of chamfer.hip (every distance is the same fma chain
// fma(dz,dz, fma(dy,dy, dx*dx)) with d = query - candidate, and ties go to the lowest candidate index), i.e. as
// pytorch3d.ops.knn_points(K=1) behind pytorch3d.loss.chamfer_distance (utility/utils.py:207,212) — but a query only
// meets the candidates that can still beat what it has:
//
//   1. nn_sort_kernel   one workgroup per cloud: bounding box, a G^3 grid over it (G = 16, 32 above 20k points), a counting
//                       sort of the points by the Hilbert-curve position of their cell (histogram and cursors in LDS).  The sorted
//                       cloud is stored as float4 (x, y, z, original index): 64 consecutive points — a "block" — are a
//                       compact patch of the surface.
//   2. nn_boxes_kernel  per block (one wave each): its axis-aligned bounding box (the lane-parallel coarse test), an ORIENTED box
//                       in the principal frame of its 64 points and one per 16-point group (below: "oriented bounding
//                       boxes"); pads the last block.
//   3. nn_query_kernel  one wave per block of 64 sorted QUERIES (one per lane, so the lanes of a wave ask about the same
//                       neighbourhood).  The wave tests the candidate blocks 64 at a time, lane-parallel, box against
//                       box; a block survives if its box-to-box lower bound does not exceed the worst "best so far" of
//                       the wave, is then tested per lane (point to oriented box), then its four groups (point to the
//                       group's oriented box), and only groups some lane can still improve on are evaluated — 16 candidates
//                       broadcast through the scalar cache, 6.5 VALU operations per pair as in the brute-force loop.
//                       Every bound is <= the COMPUTED distance to every point it covers (monotone arithmetic for the
//                       axis-aligned boxes, stated margins for the oriented ones), so pruning with "bound > best" can
//                       never drop a candidate that ties or wins.
//                       The arg-min is kept per 16-candidate group in the hot loop and resolved afterwards by rescanning
//                       that group for the lowest original index; if another group tied the minimum exactly (duplicate
//                       points), the wave rescans every block that can hold the distance — rare, and exact.
//
// Work per query grows slowly with the cloud instead of in proportion to it.  Measured (3 draws x 64 clouds, tools/chamfer_bench.py,
// profiles/r05_chamfer_search.txt; ms per Chamfer forward): the untrained network's geometry (sphere 0.25 against ellipsoids of
// 0.05-0.16, concentric) 10k points 1.17, 25k 3.12, 3 x 8 x 50k 1.63 against 3.8 / 20.0 / 12.0 for the single-sweep brute
// force; a sphere against an offset ellipsoid 0.90 / 2.23 / 1.05.  Round 4 took 2.96 / 8.09 / 8.50 and 0.93 / 2.56 / 1.74:
// what changed is (a) the oriented boxes — a sampled surface patch is ~10 x thinner along its normal than its axis-aligned
// box, and with the surfaces 0.1-0.2 apart three times as many axis-aligned boxes passed the bound — and (b) the pad lanes of a
// cloud's last query block, which asked about the origin instead of repeating lane 0's point (nn_query_wave).
// Variants that halve the scalar fetches per query (two query blocks per wave), remove them (candidates and boxes through
// vector loads + DPP row_newbcast, 17-25 % slower) or interleave clouds over CUs (persistent grid) are not faster — DESIGN.md
// §4, §8.  All kernels are deterministic in their outputs (the order of points inside a grid cell depends on LDS atomics, the
// minima do not).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace a3vt {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kPB = 64;                 // points per block = lanes per wave
constexpr float kPadCoord = 1.0e18f;    // coordinates of the pad entries of a cloud's last block (never a minimum)
constexpr int kSortThreads = 1024;

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}
// (__builtin_bit_cast applied directly to an element of an ext_vector reads element 0 with this compiler: go through
// a by-value argument)
__device__ __forceinline__ int as_int(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ float sq3(float dx, float dy, float dz) {
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));   // the distance arithmetic of chamfer.hip
}
// distance between the intervals [qlo, qhi] and [tlo, thi] (0 if they overlap); <= |q - c| for q, c inside them, also
// as computed: fp subtraction is monotone
__device__ __forceinline__ float gap(float qlo, float qhi, float tlo, float thi) {
  return fmaxf(fmaxf(tlo - qhi, qlo - thi), 0.f);
}
__device__ __forceinline__ unsigned spread3(unsigned v) {   // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// Position of grid cell (x, y, z) along the 3-D Hilbert curve of 2^bits cells per axis (Skilling's transposition).  The
// curve is continuous, so 64 consecutive sorted points form a connected patch; consecutive Morton cells jump, which made
// the blocks' boxes a quarter longer and nearly twice as large in surface (measured on 10k-point ellipsoids).
__device__ __forceinline__ unsigned hilbert3(unsigned x, unsigned y, unsigned z, int bits) {
  unsigned X[3] = {x, y, z};
  const unsigned M = 1u << (bits - 1);
  for (unsigned Q = M; Q > 1; Q >>= 1) {
    const unsigned P = Q - 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (X[i] & Q) {
        X[0] ^= P;
      } else {
        const unsigned t = (X[0] ^ X[i]) & P;
        X[0] ^= t;
        X[i] ^= t;
      }
    }
  }
  X[1] ^= X[0];
  X[2] ^= X[1];
  unsigned t = 0;
  for (unsigned Q = M; Q > 1; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1;
  return (spread3(X[0] ^ t) << 2) | (spread3(X[1] ^ t) << 1) | spread3(X[2] ^ t);
}

struct NNClouds {   // the clouds of one Chamfer call: nx predicted clouds of p points, ny ground-truth clouds of q
  const float *x, *y;
  int p, q, nx, ny;
  int npx, npy;     // padded points per cloud (multiple of 64)
  f32x4 *sx, *sy;   // sorted clouds  [n?][np?]
  f32x4 *bx, *by;   // block boxes    [n?][np?/64][2]  (min xyz 0, max xyz 0): the lane-parallel coarse test
  float *ox, *oy;   // block ORIENTED boxes [n?][np?/64][16]  (Obb below)
  float *gx, *gy;   // group oriented boxes [n?][np?/16][16]  one per 16 consecutive sorted points (4 per block)
};

// ---- oriented bounding boxes ---------------------------------------------------------------------------------------------
// A sampled surface patch is thin along its normal: the axis-aligned box of 16-64 neighbouring samples of a tilted patch is
// ~10 x thicker than the patch, and with the untrained network's surfaces 0.1-0.2 apart (patches 0.02-0.05 wide) every
// patch within ~0.1 of the nearest one passes an axis-aligned bound — 78 groups of 16 candidates evaluated per wave of 64
// queries at 10 000 points (profiles/r05_nn_pruning_stats.txt).  A box in the patch's own principal frame is as thin as the
// patch, and a third as many pass.
//
// Record (16 floats = one s_load_dwordx16): centre c[3], three unit axes a[3][3], half extents e[3], pad.
// Bound of a query q:  t_i = a_i . (q - c),  gap_i = max(|t_i| - e_i - mu, 0),  lb = (gap_0^2 + gap_1^2 + gap_2^2) (1 - 2^-16).
// lb <= the COMPUTED squared distance from q to every point p of the box's group, by construction:
//   * in exact arithmetic |a_i . (q - p)| >= |t_i(q)| - |t_i(p)| and sum_i (a_i . v)^2 <= (1 + d) |v|^2, where d < 2^-18 is
//     what the Gram-Schmidt'ed fp32 axes lack of orthonormality (checked per box: norms and dot products within 2^-20);
//   * t_i is computed by the SAME device function (obb_proj) for the points when the extents are taken and for the queries,
//     with |rounding error| <= 2^-22 |v|_1 (three subtractions, a product and two fma with |a_ij| <= 1): the extents are
//     stored inflated by 2^-20 max_p |p - c|_1 and the query pays mu = 2^-20 |q - c|_1;
//   * the factor (1 - 2^-16) covers d, the roundings of the three squares and their sum, and the 3 x 2^-24 of the distance
//     itself.
// A candidate that ties or wins therefore always has lb <= best and is never skipped (the same guarantee the monotone
// axis-aligned bounds gave, with margins instead of monotonicity).  Groups with no valid point store e = -3e38 (lb = +inf),
// groups with a non-finite coordinate store e = +3e38 (lb = 0: never pruned by this bound).
constexpr int kObbFloats = 16;
constexpr float kObbMu = 9.5367431640625e-07f;          // 2^-20
constexpr float kObbShrink = 1.0f - 1.52587890625e-05f;   // 1 - 2^-16
// (The multiplication stops shrinking once the sum of squared gaps is denormal — coordinates below ~1e-19 — where the absolute
// 1e-37 slack on the extents is what is left; tests/test_gpu_nn_pruned.py::test_oriented_boxes_at_denormal_squared_distances
// holds the search to the brute force bit for bit at coordinate scales 1e-15 .. 1e-22.)

__device__ __forceinline__ float obb_proj(float ax, float ay, float az, float dx, float dy, float dz) {
  return __builtin_fmaf(az, dz, __builtin_fmaf(ay, dy, ax * dx));
}
// o: the record as four float4 (wave-uniform: scalar registers in the query kernel)
struct Obb { f32x4 v[4]; };
__device__ __forceinline__ float obb_lb(float qx, float qy, float qz, const Obb &o) {
  const float dx = qx - o.v[0][0], dy = qy - o.v[0][1], dz = qz - o.v[0][2];
  const float mu = (__builtin_fabsf(dx) + __builtin_fabsf(dy) + __builtin_fabsf(dz)) * kObbMu;
  const float g0 = fmaxf(__builtin_fabsf(obb_proj(o.v[0][3], o.v[1][0], o.v[1][1], dx, dy, dz)) - (o.v[3][0] + mu), 0.f);
  const float g1 = fmaxf(__builtin_fabsf(obb_proj(o.v[1][2], o.v[1][3], o.v[2][0], dx, dy, dz)) - (o.v[3][1] + mu), 0.f);
  const float g2 = fmaxf(__builtin_fabsf(obb_proj(o.v[2][1], o.v[2][2], o.v[2][3], dx, dy, dz)) - (o.v[3][2] + mu), 0.f);
  return __builtin_fmaf(g2, g2, __builtin_fmaf(g1, g1, g0 * g0)) * kObbShrink;
}

// Eigenvectors of a symmetric 3 x 3 matrix by cyclic Jacobi rotations (columns of V); any orthonormal V gives a VALID box,
// the principal frame gives a thin one.
__device__ inline void eig3_jacobi(float A[3][3], float V[3][3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.f : 0.f;
  // (hardware reciprocal / square-root approximations: a rotation that is orthonormal to 1e-6 only is good enough — the frame
  // is re-orthonormalised exactly by the caller and checked; four sweeps bring a 3 x 3 matrix to fp32 convergence)
  for (int sweep = 0; sweep < 4; ++sweep) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int p = k == 2 ? 1 : 0, q = k == 0 ? 1 : 2;
      const float apq = A[p][q];
      if (!(__builtin_fabsf(apq) > 1.0e-12f * (__builtin_fabsf(A[p][p]) + __builtin_fabsf(A[q][q])))) continue;
      const float theta = (A[q][q] - A[p][p]) * __builtin_amdgcn_rcpf(2.f * apq);
      const float t = (theta >= 0.f ? 1.f : -1.f) * __builtin_amdgcn_rcpf(__builtin_fabsf(theta) + __builtin_amdgcn_sqrtf(theta * theta + 1.f));
      const float c = __builtin_amdgcn_rsqf(t * t + 1.f), sn = t * c;
      const int r = 3 - p - q;
      const float app = A[p][p], aqq = A[q][q], arp = A[r][p], arq = A[r][q];
      A[p][p] = app - t * apq;
      A[q][q] = aqq + t * apq;
      A[p][q] = A[q][p] = 0.f;
      A[r][p] = A[p][r] = c * arp - sn * arq;
      A[r][q] = A[q][r] = sn * arp + c * arq;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float vp = V[i][p], vq = V[i][q];
        V[i][p] = c * vp - sn * vq;
        V[i][q] = sn * vp + c * vq;
      }
    }
  }
}

// The oriented box of the valid points among the 2^LOG lanes that share (lane >> LOG): every lane of the group gets the record.
template <int LOG>
__device__ inline void group_obb(const f32x4 v, bool valid, float out[kObbFloats]) {
  auto gsum = [](float x) {
#pragma unroll
    for (int off = 1; off < (1 << LOG); off <<= 1) x += __shfl_xor(x, off, 64);
    return x;
  };
  auto gmax = [](float x) {
#pragma unroll
    for (int off = 1; off < (1 << LOG); off <<= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
    return x;
  };
  auto gmin = [](float x) {
#pragma unroll
    for (int off = 1; off < (1 << LOG); off <<= 1) x = fminf(x, __shfl_xor(x, off, 64));
    return x;
  };
  const float w = valid ? 1.f : 0.f;
  const float n = gsum(w);
#pragma unroll
  for (int i = 0; i < kObbFloats; ++i) out[i] = 0.f;
  out[3] = out[7] = out[11] = 1.f;
  if (!(n > 0.f)) {   // no valid point: every bound against this group is +inf
    out[12] = out[13] = out[14] = -3.0e38f;
    return;
  }
  const float inv = 1.f / n;
  float c[3], d[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    c[k] = gsum(valid ? v[k] : 0.f) * inv;
    d[k] = valid ? v[k] - c[k] : 0.f;
  }
  float A[3][3], V[3][3];
  A[0][0] = gsum(d[0] * d[0]), A[0][1] = A[1][0] = gsum(d[0] * d[1]), A[0][2] = A[2][0] = gsum(d[0] * d[2]);
  A[1][1] = gsum(d[1] * d[1]), A[1][2] = A[2][1] = gsum(d[1] * d[2]), A[2][2] = gsum(d[2] * d[2]);
  eig3_jacobi(A, V);
#ifdef A3VT_DBG_NN_AABB   // A/B build: coordinate axes, i.e. the axis-aligned boxes of rounds 2-4 in this record (what the principal frame buys)
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.f : 0.f;
#endif
  // axes = columns of V, re-orthonormalised (Gram-Schmidt; the third as the cross product of the first two)
  float a[3][3];
  {
    float l = __builtin_sqrtf(V[0][0] * V[0][0] + V[1][0] * V[1][0] + V[2][0] * V[2][0]);
    l = l > 0.f ? 1.f / l : 0.f;
    a[0][0] = V[0][0] * l, a[0][1] = V[1][0] * l, a[0][2] = V[2][0] * l;
    const float dt = a[0][0] * V[0][1] + a[0][1] * V[1][1] + a[0][2] * V[2][1];
    float b0 = V[0][1] - dt * a[0][0], b1 = V[1][1] - dt * a[0][1], b2 = V[2][1] - dt * a[0][2];
    l = __builtin_sqrtf(b0 * b0 + b1 * b1 + b2 * b2);
    l = l > 0.f ? 1.f / l : 0.f;
    a[1][0] = b0 * l, a[1][1] = b1 * l, a[1][2] = b2 * l;
    a[2][0] = a[0][1] * a[1][2] - a[0][2] * a[1][1];
    a[2][1] = a[0][2] * a[1][0] - a[0][0] * a[1][2];
    a[2][2] = a[0][0] * a[1][1] - a[0][1] * a[1][0];
  }
  // centre of the box in its own frame, then the half extents about THAT centre in the arithmetic the queries use
  float lo[3], hi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float t = obb_proj(a[k][0], a[k][1], a[k][2], d[0], d[1], d[2]);
    lo[k] = gmin(valid ? t : 3.0e38f);
    hi[k] = gmax(valid ? t : -3.0e38f);
  }
  float cc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
    cc[k] = c[k] + 0.5f * ((lo[0] + hi[0]) * a[0][k] + (lo[1] + hi[1]) * a[1][k] + (lo[2] + hi[2]) * a[2][k]);
  const float ex = valid ? v[0] - cc[0] : 0.f, ey = valid ? v[1] - cc[1] : 0.f, ez = valid ? v[2] - cc[2] : 0.f;
  const float m1 = gmax(__builtin_fabsf(ex) + __builtin_fabsf(ey) + __builtin_fabsf(ez));
  float e[3];
  bool finite = m1 < 3.0e38f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float t = gmax(__builtin_fabsf(obb_proj(a[k][0], a[k][1], a[k][2], ex, ey, ez)));
    e[k] = t * (1.f + kObbMu) + kObbMu * m1 + 1.0e-37f;
    finite = finite && e[k] < 3.0e38f && __builtin_fabsf(cc[k]) < 3.0e38f;
#pragma unroll
    for (int j = 0; j < 3; ++j) finite = finite && __builtin_fabsf(a[k][j]) <= 1.0001f;
  }
  // (the axes must be orthonormal to 2^-18 for the bound: a frame that is not — zero vectors from a degenerate rotation,
  // NaN — is replaced by the coordinate frame with unbounded extents)
  const float n0 = a[0][0] * a[0][0] + a[0][1] * a[0][1] + a[0][2] * a[0][2];
  const float n1 = a[1][0] * a[1][0] + a[1][1] * a[1][1] + a[1][2] * a[1][2];
  const float n2 = a[2][0] * a[2][0] + a[2][1] * a[2][1] + a[2][2] * a[2][2];
  const float d01 = a[0][0] * a[1][0] + a[0][1] * a[1][1] + a[0][2] * a[1][2];
  const float d02 = a[0][0] * a[2][0] + a[0][1] * a[2][1] + a[0][2] * a[2][2];
  const float d12 = a[1][0] * a[2][0] + a[1][1] * a[2][1] + a[1][2] * a[2][2];
  const float tol = 9.5367431640625e-07f;   // 2^-20
  finite = finite && __builtin_fabsf(n0 - 1.f) <= tol && __builtin_fabsf(n1 - 1.f) <= tol && __builtin_fabsf(n2 - 1.f) <= tol &&
           __builtin_fabsf(d01) <= tol && __builtin_fabsf(d02) <= tol && __builtin_fabsf(d12) <= tol;
  if (!finite) {   // never pruned by this bound (the axis-aligned coarse test and the distances themselves still decide)
    out[12] = out[13] = out[14] = 3.0e38f;
    return;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    out[k] = cc[k];
    out[12 + k] = e[k];
#pragma unroll
    for (int j = 0; j < 3; ++j) out[3 + 3 * k + j] = a[k][j];
  }
}

__host__ __device__ inline int nn_grid_bits(int n) { return n > 20000 ? 5 : 4; }

__global__ __launch_bounds__(kSortThreads) void nn_sort_kernel(NNClouds c) {
  extern __shared__ unsigned hist[];   // G^3 counters, then cursors
  __shared__ float red[6][16];
  __shared__ unsigned wsum[16];
  const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool isx = cloud < c.nx;
  const int n = isx ? c.p : c.q;
  const float *P = isx ? c.x + (size_t)cloud * c.p * 3 : c.y + (size_t)(cloud - c.nx) * c.q * 3;
  f32x4 *out = isx ? c.sx + (size_t)cloud * c.npx : c.sy + (size_t)(cloud - c.nx) * c.npy;
  const int bits = nn_grid_bits(n), G = 1 << bits, cells = G * G * G;

  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int i = tid; i < n; i += kSortThreads) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float v = P[i * 3 + d];
      mn[d] = fminf(mn[d], v);
      mx[d] = fmaxf(mx[d], v);
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    mn[d] = wave_min(mn[d]);
    mx[d] = wave_max(mx[d]);
  }
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) red[d][wave] = mn[d], red[3 + d][wave] = mx[d];
  }
  for (int i = tid; i < cells; i += kSortThreads) hist[i] = 0;
  __syncthreads();
  float scale[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float a = red[d][0], b = red[3 + d][0];
#pragma unroll
    for (int w = 1; w < 16; ++w) a = fminf(a, red[d][w]), b = fmaxf(b, red[3 + d][w]);
    mn[d] = a;
    const float ext = b - a;
    scale[d] = (ext > 0.f && ext < 3.0e38f) ? (float)G / ext : 0.f;   // degenerate / non-finite extent: one cell
  }
  auto key_of = [&](int i) {
    unsigned cell[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float t = fminf(fmaxf((P[i * 3 + d] - mn[d]) * scale[d], 0.f), (float)(G - 1));   // NaN -> 0
      cell[d] = (unsigned)(int)t;
    }
    return hilbert3(cell[0], cell[1], cell[2], bits);
  };
  for (int i = tid; i < n; i += kSortThreads) atomicAdd(&hist[key_of(i)], 1u);
  __syncthreads();
  // exclusive scan of the counters: `per` consecutive cells per thread, then a scan of the 1024 partial sums
  const int per = cells / kSortThreads;   // 4 or 32
  unsigned local = 0;
  for (int k = 0; k < per; ++k) local += hist[tid * per + k];
  unsigned incl = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  unsigned base = incl - local;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  for (int k = 0; k < per; ++k) {
    const unsigned t = hist[tid * per + k];
    hist[tid * per + k] = base;
    base += t;
  }
  __syncthreads();
  for (int i = tid; i < n; i += kSortThreads) {
    const unsigned pos = atomicAdd(&hist[key_of(i)], 1u);
    out[pos] = f32x4{P[i * 3 + 0], P[i * 3 + 1], P[i * 3 + 2], __builtin_bit_cast(float, i)};
  }
}

// grid = ceil(blocks / 16) x clouds (flattened: gridDim.y stops at 65 535 clouds), 16 waves per workgroup, one block of
// 64 sorted points per wave
__global__ __launch_bounds__(1024) void nn_boxes_kernel(NNClouds c, int wgs_per_cloud) {
  const int cloud = blockIdx.x / wgs_per_cloud, bx = blockIdx.x % wgs_per_cloud;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool isx = cloud < c.nx;
  const int n = isx ? c.p : c.q, np = isx ? c.npx : c.npy;
  const int blk = bx * 16 + wave;
  if (blk * kPB >= np) return;
  f32x4 *pts = isx ? c.sx + (size_t)cloud * c.npx : c.sy + (size_t)(cloud - c.nx) * c.npy;
  f32x4 *box = (isx ? c.bx + (size_t)cloud * (c.npx / kPB) * 2 : c.by + (size_t)(cloud - c.nx) * (c.npy / kPB) * 2) + blk * 2;
  const int i = blk * kPB + lane;
  const bool valid = i < n;
  f32x4 v = {kPadCoord, kPadCoord, kPadCoord, __builtin_bit_cast(float, -1)};
  if (valid) v = pts[i];
  else pts[i] = v;
  float lo[3], hi[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    lo[d] = valid ? v[d] : 3.0e38f;
    hi[d] = valid ? v[d] : -3.0e38f;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {   // the 16 points of this lane's group
      lo[d] = fminf(lo[d], __shfl_xor(lo[d], off, 64));
      hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], off, 64));
    }
  }
  // oriented boxes: one per 16-point group (every lane of a group holds its record; lane 0 of the group stores it), one for
  // the block
  {
    float o[kObbFloats];
    group_obb<4>(v, valid, o);
    float *go = (isx ? c.gx + (size_t)cloud * (c.npx / 16) * kObbFloats : c.gy + (size_t)(cloud - c.nx) * (c.npy / 16) * kObbFloats) +
                (size_t)(blk * 4 + (lane >> 4)) * kObbFloats;
    if ((lane & 15) < 4) {
      const int k = lane & 15;
      reinterpret_cast<f32x4 *>(go)[k] = k == 0 ? f32x4{o[0], o[1], o[2], o[3]} : k == 1 ? f32x4{o[4], o[5], o[6], o[7]}
                                       : k == 2 ? f32x4{o[8], o[9], o[10], o[11]} : f32x4{o[12], o[13], o[14], o[15]};
    }
    group_obb<6>(v, valid, o);
    float *bo = (isx ? c.ox + (size_t)cloud * (c.npx / kPB) * kObbFloats : c.oy + (size_t)(cloud - c.nx) * (c.npy / kPB) * kObbFloats) +
                (size_t)blk * kObbFloats;
    if (lane < 4) {
      const int k = lane;
      reinterpret_cast<f32x4 *>(bo)[k] = k == 0 ? f32x4{o[0], o[1], o[2], o[3]} : k == 1 ? f32x4{o[4], o[5], o[6], o[7]}
                                       : k == 2 ? f32x4{o[8], o[9], o[10], o[11]} : f32x4{o[12], o[13], o[14], o[15]};
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
      lo[d] = fminf(lo[d], __shfl_xor(lo[d], off, 64));
      hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], off, 64));
    }
  }
  if (lane == 0) {
    box[0] = f32x4{lo[0], lo[1], lo[2], 0.f};
    box[1] = f32x4{hi[0], hi[1], hi[2], 0.f};
  }
}

#ifdef A3VT_DBG_NN_TRACE   // per-wave timeline only (tools/nn_trace.py): two s_memrealtime reads and one record per wave
constexpr int kNNTraceWaves = 1 << 16;
__device__ unsigned long long nn_trace[kNNTraceWaves][4];   // per wave: start, end (s_memrealtime, 100 MHz), pair / block, groups << 32 | tests
__device__ unsigned nn_trace_n;
#endif
#ifdef A3VT_DBG_NN_STATS   // developer counters (tools/build_variants.sh nn): waves, blocks evaluated, point-box tests, slow paths
__device__ unsigned long long nn_inflight;
__device__ unsigned long long nn_stats[16];   // [8..13]: shader-clock cycles per phase (whole wave, seed search, tests, evaluations, tail)
#define NN_STAT(i, v) do { if (lane == 0) atomicAdd(&nn_stats[i], (unsigned long long)(v)); } while (0)
#define NN_CLOCK() __builtin_readcyclecounter()
#else
#define NN_STAT(i, v) do { } while (0)
#define NN_CLOCK() 0ull
#endif

// Wave-uniform maximum / minimum of NON-NEGATIVE floats through their bit patterns (they order like unsigned integers):
// four DPP steps inside each 16-lane row, then the four row results through v_readlane and scalar min / max — a dozen
// instructions and no LDS round trip, where six __shfl_xor steps cost six dependent ds_bpermute.
template <int CTRL>
__device__ __forceinline__ unsigned nn_dpp(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ float wave_umax(float v) {
  unsigned x = __builtin_bit_cast(unsigned, v);
  x = max(x, nn_dpp<0xB1>(x));    // quad_perm [1,0,3,2]
  x = max(x, nn_dpp<0x4E>(x));    // quad_perm [2,3,0,1]
  x = max(x, nn_dpp<0x124>(x));   // row_ror:4
  x = max(x, nn_dpp<0x128>(x));   // row_ror:8
  const unsigned a = __builtin_amdgcn_readlane((int)x, 0), b = __builtin_amdgcn_readlane((int)x, 16),
                 c = __builtin_amdgcn_readlane((int)x, 32), d = __builtin_amdgcn_readlane((int)x, 48);
  return __builtin_bit_cast(float, max(max(a, b), max(c, d)));
}
__device__ __forceinline__ float wave_umin(float v) {
  unsigned x = __builtin_bit_cast(unsigned, v);
  x = min(x, nn_dpp<0xB1>(x));
  x = min(x, nn_dpp<0x4E>(x));
  x = min(x, nn_dpp<0x124>(x));
  x = min(x, nn_dpp<0x128>(x));
  const unsigned a = __builtin_amdgcn_readlane((int)x, 0), b = __builtin_amdgcn_readlane((int)x, 16),
                 c = __builtin_amdgcn_readlane((int)x, 32), d = __builtin_amdgcn_readlane((int)x, 48);
  return __builtin_bit_cast(float, min(min(a, b), min(c, d)));
}

struct NNQuery {
  const f32x4 *sx, *sy, *bx, *by;
  const float *ox, *oy, *gx, *gy;   // oriented boxes of the blocks / of the 16-point groups
  int p, q, npx, npy, nz, batch;   // nz = draws * batch cloud pairs; pair z = (x cloud z, y cloud z % batch)
  float *dxy, *dyx;
  int32_t *ixy, *iyx;
  unsigned long long *work;   // a3vt_dbg_nn_work: nullptr (the product: no counter is touched) or 8 counters
};
// Work counters of the search for regression tests (a3vt_dbg_nn_work): [0] waves, [1] groups of 16 candidates evaluated,
// [2] blocks evaluated, [3] point-to-box tests, [4] most groups evaluated by one wave.  Off unless a test turns them on.
__device__ unsigned long long g_nn_work[8];
static bool g_nn_work_on = false;

// One wave = one block of 64 sorted queries.  y < nz: the x cloud asks the y cloud; y >= nz: the y cloud asks the x cloud.
__device__ __forceinline__ void nn_query_wave(const NNQuery &a, int y, int qblk, int lane) {
  const bool fwd = y < a.nz;
  const int z = fwd ? y : y - a.nz;
  const int zx = z, zy = z % a.batch;
  const int nbx = a.npx / kPB, nby = a.npy / kPB;
  const int nqb = fwd ? nbx : nby, ntb = fwd ? nby : nbx;
  if (qblk >= nqb) return;
  const f32x4 *__restrict__ qpts = fwd ? a.sx + (size_t)zx * a.npx : a.sy + (size_t)zy * a.npy;
  const f32x4 *__restrict__ qbox = fwd ? a.bx + (size_t)zx * nbx * 2 : a.by + (size_t)zy * nby * 2;
  const f32x4 *__restrict__ tpts = fwd ? a.sy + (size_t)zy * a.npy : a.sx + (size_t)zx * a.npx;
  const f32x4 *__restrict__ tbox = fwd ? a.by + (size_t)zy * nby * 2 : a.bx + (size_t)zx * nbx * 2;
  const float *__restrict__ tobb = fwd ? a.oy + (size_t)zy * nby * kObbFloats : a.ox + (size_t)zx * nbx * kObbFloats;
  const float *__restrict__ tgrp = fwd ? a.gy + (size_t)zy * nby * 4 * kObbFloats : a.gx + (size_t)zx * nbx * 4 * kObbFloats;
  float *od = fwd ? a.dxy + (size_t)z * a.p : a.dyx + (size_t)z * a.q;
  int32_t *oi = fwd ? a.ixy + (size_t)z * a.p : a.iyx + (size_t)z * a.q;

  const f32x4 me = qpts[(size_t)qblk * kPB + lane];
  const int qidx = as_int(me[3]);
  // Pad lanes (last block of the cloud) ask for lane 0's point and write nothing.  Lane 0's coordinates are read with
  // v_readlane BEFORE the select: as `qidx < 0 ? __shfl(me[0], 0, 64) : me[0]` the shuffle ran in the branch of the pad lanes
  // only, where lane 0 is inactive — a bpermute from an inactive lane returns 0, so the pad lanes asked about the ORIGIN.
  // Harmless for the results (they write nothing) and ruinous for the time whenever the origin is the centre of the target
  // (the untrained network's sphere): everything is then equidistant, nothing can be pruned, and the last query block of
  // every cloud evaluated 440-590 of the 625 groups — 192 waves of 2.2 ms each at the tail of a 1.5 ms launch
  // (profiles/r05_nn_wave_timeline.txt).
  const float q0x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(as_int(me[0]), 0));
  const float q0y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(as_int(me[1]), 0));
  const float q0z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(as_int(me[2]), 0));
  const float qx = qidx < 0 ? q0x : me[0];
  const float qy = qidx < 0 ? q0y : me[1];
  const float qz = qidx < 0 ? q0z : me[2];
  const f32x4 qb0 = qbox[qblk * 2], qb1 = qbox[qblk * 2 + 1];

  const unsigned long long clk0 = NN_CLOCK();
  unsigned long long clk_test = 0, clk_eval = 0, clk_ld = 0, clk_bd = 0, clk_gr = 0;
#ifdef A3VT_DBG_NN_STATS
  if (lane == 0) clk_ld = atomicAdd(&nn_inflight, 1ull);   // waves of this kernel in flight when this one starts
#endif
#ifdef A3VT_DBG_NN_TRACE
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  float best = 3.0e38f;
  int bsub = 0;        // 16-candidate group (block * 4 + quarter) that gave `best`
  bool tie = false;    // another group reproduced `best` exactly
  int n_grp = 0;       // (developer counter) 16-candidate groups evaluated
  // Wave-uniform data (a candidate block's oriented boxes, its 64 points) comes through the scalar cache: every batch of
  // loads is issued back to back and waited for once (scalar loads return out of order).  (Delivering it through the vector
  // path instead — each lane loads element lane & 15 of a record / one candidate of a group, DPP row broadcasts folded into
  // the consuming v_sub / v_mul hand element k to the row at no instruction cost — is 17-25 % SLOWER at every size:
  // tools/experiments/nn_query_vector_path_r05.patch)
  auto eval = [&](int blk) {
    // the oriented boxes of the block's four groups: 4 x s_load_dwordx16
    const f32x4 *__restrict__ sg = reinterpret_cast<const f32x4 *>(tgrp) + (size_t)blk * 16;
    float lbs[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {   // the group's own box: a quarter of the block's points sit in a box far thinner than the block's
      Obb o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o.v[k] = sg[s * 4 + k];
      lbs[s] = obb_lb(qx, qy, qz, o);
    }
    // (the four records are 64 scalar registers, a group's 16 candidates another 64: the candidate block index is redefined
    // behind the four bounds, so that no candidate load is hoisted into the records' live range — that spilled 105 scalar
    // registers into vector lanes.  Not volatile: a volatile asm counts as a store, and loads behind a store cannot use the
    // scalar cache; readfirstlane because the compiler takes an asm result for divergent.)
    int blk_c = blk;
    asm("" : "+s"(blk_c) : "v"(lbs[0]), "v"(lbs[1]), "v"(lbs[2]), "v"(lbs[3]));
    blk_c = __builtin_amdgcn_readfirstlane(blk_c);
    const f32x4 *__restrict__ tp = tpts + (size_t)blk_c * kPB;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (__builtin_amdgcn_ballot_w64(lbs[s] <= best) == 0) continue;
      ++n_grp;
      f32x4 cnd[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) cnd[j] = tp[s * 16 + j];
      float m = 3.0e38f;
#pragma unroll
      for (int j = 0; j < 16; j += 4) {
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = sq3(qx - cnd[j + u][0], qy - cnd[j + u][1], qz - cnd[j + u][2]);
        m = __builtin_fminf(__builtin_fminf(m, d[0]), d[1]);
        m = __builtin_fminf(__builtin_fminf(m, d[2]), d[3]);
      }
      if (m < best) {
        best = m;
        bsub = blk * 4 + s;
        tie = false;
      } else if (m == best) {
        tie = true;
      }
    }
  };
  auto point_box = [&](int blk) {   // lower bound of this lane's distance to any point of candidate block blk (uniform)
    Obb o;
    const f32x4 *__restrict__ sb = reinterpret_cast<const f32x4 *>(tobb) + (size_t)blk * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) o.v[i] = sb[i];
    return obb_lb(qx, qy, qz, o);
  };
  auto box_box = [&](int blk) {     // per lane: lower bound between the query block's box and candidate block blk's
    const f32x4 t0 = tbox[blk * 2], t1 = tbox[blk * 2 + 1];
    return sq3(gap(qb0[0], qb1[0], t0[0], t1[0]), gap(qb0[1], qb1[1], t0[1], t1[1]), gap(qb0[2], qb1[2], t0[2], t1[2]));
  };

  // seed: the candidate block nearest to the query block, box to box (lowest index among equals)
  float sb = 3.0e38f;
  int sblk = 0x7fffffff;
  for (int it = 0; it < ntb; it += 64) {
    const int b = it + lane;
    if (b < ntb) {
      const float l = box_box(b);
      if (l < sb) sb = l, sblk = b;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(sb, off, 64);
    const int ob = __shfl_xor(sblk, off, 64);
    if (o < sb || (o == sb && ob < sblk)) sb = o, sblk = ob;
  }
  const int seed = min(__builtin_amdgcn_readfirstlane(sblk), ntb - 1);
  int n_eval = 1, n_test = 0;
  const unsigned long long clk1 = NN_CLOCK();
  eval(seed);
  float T = wave_umax(best);
  const unsigned long long clk2 = NN_CLOCK();   // (wave-uniform) a block whose box-to-box bound exceeds this cannot help any lane

  for (int it = 0; it < ntb; it += 64) {
    const int b = it + lane;
    const float lbb = b < ntb ? box_box(b) : 3.0e38f;
    unsigned long long mask = __builtin_amdgcn_ballot_w64(lbb <= T && b < ntb && b != seed);
    while (mask) {
      // nearest surviving block first (box to box): the minima tighten early and the later blocks fail their tests
      const unsigned long long ca = NN_CLOCK();
      const float mine = ((mask >> lane) & 1ull) ? lbb : 3.0e38f;
      const float nearest = wave_umin(mine);
      const unsigned long long pick = __builtin_amdgcn_ballot_w64(mine == nearest) & mask;
      const int bit = __builtin_ctzll(pick ? pick : mask);
      mask &= ~(1ull << bit);
      const int blk = it + bit;
      const float lb = point_box(blk);
      ++n_test;
      const bool skip = __builtin_amdgcn_ballot_w64(lb <= best) == 0;   // <=: a candidate that TIES must still be seen
      const unsigned long long cb = NN_CLOCK();
      clk_test += cb - ca;
      if (skip) continue;
      eval(blk);
      ++n_eval;
      T = wave_umax(best);
      mask &= __builtin_amdgcn_ballot_w64(lbb <= T);
      clk_eval += NN_CLOCK() - cb;
    }
  }

  // lowest original index among the candidates of the winning group that reproduce the minimum
  int bidx = 0x7fffffff;
  {
    const f32x4 *rp = tpts + (size_t)bsub * 16;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      const f32x4 cnd = rp[j];
      const int ci = as_int(cnd[3]);
      if (sq3(qx - cnd[0], qy - cnd[1], qz - cnd[2]) == best && ci >= 0) bidx = min(bidx, ci);
    }
  }
  if (a.work != nullptr && lane == 0) {
    atomicAdd(&a.work[0], 1ull);
    atomicAdd(&a.work[1], (unsigned long long)n_grp);
    atomicAdd(&a.work[2], (unsigned long long)n_eval);
    atomicAdd(&a.work[3], (unsigned long long)n_test);
    atomicMax(&a.work[4], (unsigned long long)n_grp);
  }
  const unsigned long long clk3 = NN_CLOCK();
  NN_STAT(0, 1);
  NN_STAT(1, n_eval);
  NN_STAT(2, n_test);
  NN_STAT(5, n_grp);
#ifdef A3VT_DBG_NN_STATS
  if (lane == 0) atomicMax(&nn_stats[6], (unsigned long long)n_grp);
#endif
#ifdef A3VT_DBG_NN_STATS
  {   // blocks that HAD to be evaluated given the final minima (slot 3), blocks some lane needs on average (slot 2 reused below)
    int need = 0, lane_need = 0;
    for (int blk = 0; blk < ntb; ++blk) {
      const float lb = point_box(blk);
      need += __builtin_amdgcn_ballot_w64(lb <= best) != 0;
      lane_need += lb <= best;
    }
    NN_STAT(3, need);
    if (lane == 0) atomicAdd(&nn_stats[2], 0ull);
    lane_need = (int)wave_sum((float)lane_need);
    if (lane == 0) atomicAdd(&nn_stats[4], (unsigned long long)lane_need);
  }
#endif
  if (__builtin_amdgcn_ballot_w64(tie) != 0) {   // duplicates of the minimum in other groups: look everywhere it can be
    for (int blk = 0; blk < ntb; ++blk) {
      const float lb = point_box(blk);
      if (__builtin_amdgcn_ballot_w64(tie && lb <= best) == 0) continue;
      const f32x4 *__restrict__ tp = tpts + (size_t)blk * kPB;
      for (int j = 0; j < kPB; ++j) {
        const f32x4 cnd = tp[j];
        const int ci = as_int(cnd[3]);
        if (tie && ci >= 0 && sq3(qx - cnd[0], qy - cnd[1], qz - cnd[2]) == best) bidx = min(bidx, ci);
      }
    }
  }
  if (bidx == 0x7fffffff) bidx = 0;   // only with non-finite coordinates (no distance ever compared equal)
  if (qidx >= 0) {
    od[qidx] = best;
    oi[qidx] = bidx;
  }
#ifdef A3VT_DBG_NN_STATS
  if (lane == 0) {
    atomicMax(&nn_stats[13], clk3 - clk0);                     // slowest wave
  }
#endif
#ifdef A3VT_DBG_NN_STATS
  if (lane == 0) atomicAdd(&nn_inflight, ~0ull);
#endif
#ifdef A3VT_DBG_NN_TRACE
  if (lane == 0) {
    const unsigned slot = atomicAdd(&nn_trace_n, 1u);
    if (slot < kNNTraceWaves) {
      nn_trace[slot][0] = rt0;
      nn_trace[slot][1] = __builtin_amdgcn_s_memrealtime();
      nn_trace[slot][2] = (unsigned long long)y << 16 | (unsigned)(qblk & 0xffff);
      nn_trace[slot][3] = ((unsigned long long)n_grp << 32) | (unsigned)n_test;
    }
  }
#endif
  NN_STAT(8, clk3 - clk0);
  NN_STAT(14, clk_ld);
  NN_STAT(15, clk_bd);
  NN_STAT(7, clk_gr);
  NN_STAT(9, clk1 - clk0);
  NN_STAT(10, clk2 - clk1);
  NN_STAT(11, clk_test);
  NN_STAT(12, clk_eval);
}

// grid = ceil(query blocks / waves per workgroup) x 2 nz, flattened (the query blocks of one cloud pair are consecutive
// workgroups: they run at about the same time and share the candidate cloud in L2 and in the scalar caches)
__global__ __launch_bounds__(1024) void nn_query_kernel(NNQuery a, int wgs_per_pair) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one), each with its own 4 MiB L2.  In launch
  // order every cloud pair's query blocks would be spread over all eight L2s — each of them then needs the sorted points and
  // boxes of EVERY pair in flight (~40 pairs x 0.2-1 MB) and misses to the Infinity Cache on most candidate fetches.  The
  // bijective remap below hands every XCD a contiguous eighth of the (pair, query block) list: a pair's workgroups share ONE
  // L2 and ~5 pairs are live per XCD.  (Placement is a speed matter only; results do not depend on it.)
  const int nwg = gridDim.x, xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const int y = wg / wgs_per_pair, bx = wg % wgs_per_pair;
  nn_query_wave(a, y, bx * (blockDim.x >> 6) + wave, lane);
}

static NNClouds nn_layout(const float *x, const float *y, int draws, int batch, int y_batch, int p, int q, void *ws) {
  NNClouds c{};
  c.x = x, c.y = y, c.p = p, c.q = q, c.nx = draws * batch, c.ny = y_batch;
  c.npx = cdiv(p, kPB) * kPB, c.npy = cdiv(q, kPB) * kPB;
  char *w = static_cast<char *>(ws);
  c.sx = reinterpret_cast<f32x4 *>(w);
  w += (size_t)c.nx * c.npx * sizeof(f32x4);
  c.sy = reinterpret_cast<f32x4 *>(w);
  w += (size_t)c.ny * c.npy * sizeof(f32x4);
  c.bx = reinterpret_cast<f32x4 *>(w);
  w += (size_t)c.nx * (c.npx / kPB) * 2 * sizeof(f32x4);
  c.by = reinterpret_cast<f32x4 *>(w);
  w += (size_t)c.ny * (c.npy / kPB) * 2 * sizeof(f32x4);
  c.ox = reinterpret_cast<float *>(w);
  w += (size_t)c.nx * (c.npx / kPB) * kObbFloats * sizeof(float);
  c.oy = reinterpret_cast<float *>(w);
  w += (size_t)c.ny * (c.npy / kPB) * kObbFloats * sizeof(float);
  c.gx = reinterpret_cast<float *>(w);
  w += (size_t)c.nx * (c.npx / 16) * kObbFloats * sizeof(float);
  c.gy = reinterpret_cast<float *>(w);
  return c;
}

size_t nn_pruned_workspace_bytes(int draws, int batch, int p, int q) {
  const size_t nx = (size_t)draws * batch, ny = batch;
  const size_t npx = (size_t)cdiv(p, kPB) * kPB, npy = (size_t)cdiv(q, kPB) * kPB;
  return (nx * npx + ny * npy) * sizeof(f32x4) + (nx * (npx / kPB) + ny * (npy / kPB)) * 2 * sizeof(f32x4) +
         (nx * (npx / kPB) + ny * (npy / kPB) + nx * (npx / 16) + ny * (npy / 16)) * kObbFloats * sizeof(float);
}

int launch_nn_pruned(const float *x, const float *y, int draws, int batch, int p, int q, float *dxy, int32_t *ixy,
                     float *dyx, int32_t *iyx, void *ws, hipStream_t s, int y_batch) {
  if (y_batch <= 0) y_batch = batch;   // clouds in y: pair (draw, mesh b) asks y[b % y_batch]
  if (reinterpret_cast<uintptr_t>(ws) & 15) {
    set_error("chamfer_fwd: the workspace of the pruned search must be 16-byte aligned");
    return -1;
  }
  const NNClouds c = nn_layout(x, y, draws, batch, y_batch, p, q, ws);
  const int bits = nn_grid_bits(p > q ? p : q);
  const size_t shmem = (size_t)(1 << (3 * bits)) * sizeof(unsigned);
  static OncePerDevice once;
  once.run([] {
    (void)hipFuncSetAttribute((const void *)nn_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (1 << 15) * 4);
  });
#ifdef A3VT_DBG_NN_STATS   // diagnostic build only (tools/build_variants.sh nn): stop after the sort / box kernels — the outputs
  static const int stages = getenv("A3VT_NN_STAGES") ? atoi(getenv("A3VT_NN_STAGES")) : 3;   // are then NOT written
#else
  constexpr int stages = 3;
#endif
  A3VT_LAUNCH(nn_sort_kernel, dim3(c.nx + c.ny), dim3(kSortThreads), shmem, s, c);
  A3VT_CHECK_LAUNCH();
  if (stages < 2) return 0;
  const int nbmax = (c.npx > c.npy ? c.npx : c.npy) / kPB;
  const long long box_wgs = (long long)cdiv(nbmax, 16) * (c.nx + c.ny);
  const int wg_waves_q = 4;
  const long long query_wgs = (long long)cdiv(nbmax, wg_waves_q) * 2 * c.nx;
  if (box_wgs >= (1ll << 31) || query_wgs >= (1ll << 31)) {
    set_error("chamfer_fwd: %d x %d clouds of %d / %d points exceed the grid of the pruned search", draws, batch, p, q);
    return -1;
  }
  A3VT_LAUNCH(nn_boxes_kernel, dim3((unsigned)box_wgs), dim3(1024), 0, s, c, cdiv(nbmax, 16));
  A3VT_CHECK_LAUNCH();
  if (stages < 3) return 0;
  unsigned long long *work = nullptr;
  if (g_nn_work_on) {
    void *sym = nullptr;
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_nn_work)) == hipSuccess) work = static_cast<unsigned long long *>(sym);
  }
  NNQuery a{c.sx, c.sy, c.bx, c.by, c.ox, c.oy, c.gx, c.gy, p, q, c.npx, c.npy, c.nx, y_batch, dxy, dyx, ixy, iyx, work};
  A3VT_LAUNCH(nn_query_kernel, dim3((unsigned)query_wgs), dim3(64 * wg_waves_q), 0, s, a, cdiv(nbmax, wg_waves_q));
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace a3vt

#ifdef A3VT_DBG_NN_STATS
#endif
#ifdef A3VT_DBG_NN_TRACE
extern "C" int a3vt_dbg_nn_trace(unsigned long long *out, unsigned *n) {   // per-wave records of the last query launches; clears
  unsigned zero = 0;
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(a3vt::nn_trace_n), sizeof(unsigned)) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(a3vt::nn_trace), sizeof(unsigned long long) * 4 * a3vt::kNNTraceWaves) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(a3vt::nn_trace_n), &zero, sizeof(unsigned)) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int a3vt_dbg_nn_work(int enable, unsigned long long *out8) {
  // Synchronises the device (a test hook).  out8 != NULL: receives the counters accumulated so far; enable != 0: counters
  // cleared and switched on for the searches that follow, enable == 0: switched off.
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (out8 != nullptr && hipMemcpyFromSymbol(out8, HIP_SYMBOL(a3vt::g_nn_work), sizeof(z)) != hipSuccess) return -1;
  if (enable && hipMemcpyToSymbol(HIP_SYMBOL(a3vt::g_nn_work), z, sizeof(z)) != hipSuccess) return -1;
  a3vt::g_nn_work_on = enable != 0;
  return 0;
}
#ifdef A3VT_DBG_NN_STATS
extern "C" int a3vt_dbg_nn_stats(unsigned long long *out5) {   // reads and clears the counters
  unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out5, HIP_SYMBOL(a3vt::nn_stats), sizeof(z)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(a3vt::nn_stats), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
