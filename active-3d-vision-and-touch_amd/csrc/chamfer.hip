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
#ifdef A3VT_DBG_ENV   // variant builds only (tools/build_variants.sh env): the shipped library reads no environment variable
  static const int env_r = getenv("A3VT_NN_R") ? atoi(getenv("A3VT_NN_R")) : 0;  // developer override
#else
  constexpr int env_r = 0;
#endif
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

// ------------------------------------------------------------------------------------------------
// Single pass, both directions.  The two nn_kernel launches evaluate the same P x Q distance matrix twice (row minima,
// then column minima with the roles swapped).  nn2_kernel evaluates every pair ONCE: the row minima stay in registers
// exactly as above, and the column minimum of each candidate is folded on the way —
//   in the lane: min over its R queries;
//   in the wave, 4 candidates at a time: a transposing butterfly over the quad (DPP quad_perm; afterwards lane l holds
//   candidate l & 3), row_ror 4 / 8 across the quads of a 16-lane row, two xor shuffles across the rows; then the first
//   lane whose own minimum equals the wave's is found with a ballot;
//   in the workgroup: a 64-bit LDS atomic min of (distance bits << 32 | wave << 6 | lane) per candidate — non-negative
//   floats order like their bit patterns, and among equal distances the smallest (wave, lane) is the smallest query index
//   because every lane owns R CONSECUTIVE queries; per tile one 64-bit global atomic min per candidate and workgroup.
// nn2_cols_kernel then turns each candidate's (distance, wave, lane) into the index: it re-evaluates that lane's R
// queries (same arithmetic) and takes the first that reproduces the distance.  ~42 instead of 65 wave-instructions per
// candidate and 64 R queries.  Needs 8 bytes per candidate of scratch (a3vt_chamfer_scratch_bytes).
// ------------------------------------------------------------------------------------------------
constexpr int kNN2Tile = 1024;

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned x) {  // `old` = the identity of min: lets the DPP move fold into v_min_u32
  return (unsigned)__builtin_amdgcn_update_dpp(-1, (int)x, CTRL, 0xF, 0xF, false);
}

template <int R>
__global__ __launch_bounds__(256) void nn2_kernel(const float *__restrict__ xs, int np, const float *__restrict__ ys,
                                                  int nq, int batch, float *__restrict__ dxy,
                                                  int32_t *__restrict__ ixy, unsigned long long *__restrict__ colmin) {
  __shared__ __attribute__((aligned(16))) float sx[kNN2Tile + 4], sy[kNN2Tile + 4], sz[kNN2Tile + 4];
  __shared__ unsigned long long scol[kNN2Tile];
  const int z = blockIdx.y;
  const float *qb = xs + (long long)z * np * 3;
  const float *cb = ys + (long long)(z % batch) * nq * 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q0 = blockIdx.x * (256 * R) + wave * (64 * R) + lane * R;  // R consecutive queries per lane
  const unsigned gwl = ((unsigned)(blockIdx.x * 4 + wave) << 6);       // (wave of the cloud) << 6; the lane goes below

  float qx[R], qy[R], qz[R], best[R];
  int bchunk[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int qi = min(q0 + r, np - 1);
    qx[r] = qb[qi * 3 + 0];
    qy[r] = qb[qi * 3 + 1];
    qz[r] = qb[qi * 3 + 2];
    best[r] = 3.0e38f;
    bchunk[r] = 0;
  }
  const bool b0 = lane & 1, b1 = lane & 2;

  for (int t0 = 0; t0 < nq; t0 += kNN2Tile) {
    __syncthreads();
    for (int i = threadIdx.x; i < kNN2Tile; i += 256) {
      const int ci = t0 + i;
      const bool ok = ci < nq;
      sx[i] = ok ? cb[ci * 3 + 0] : kFar;
      sy[i] = ok ? cb[ci * 3 + 1] : kFar;
      sz[i] = ok ? cb[ci * 3 + 2] : kFar;
      scol[i] = ~0ull;
    }
    __syncthreads();
    const int ntile = min(kNN2Tile, nq - t0);
    const int nchunk = (ntile + kNNChunk - 1) / kNNChunk;
    for (int ch = 0; ch < nchunk; ++ch) {
      float m[R];
#pragma unroll
      for (int r = 0; r < R; ++r) m[r] = 3.0e38f;
      const int j0 = ch * kNNChunk;
      // the next four candidates are fetched while the current four are worked on (the tile arrays are padded by one
      // group, so the read past the last chunk is harmless)
      f32x4 nx = *reinterpret_cast<const f32x4 *>(sx + j0), ny = *reinterpret_cast<const f32x4 *>(sy + j0),
            nz = *reinterpret_cast<const f32x4 *>(sz + j0);
#pragma unroll 1
      for (int j = 0; j < kNNChunk; j += 4) {
        const f32x4 cx = nx, cy = ny, cz = nz;
        nx = *reinterpret_cast<const f32x4 *>(sx + j0 + j + 4);
        ny = *reinterpret_cast<const f32x4 *>(sy + j0 + j + 4);
        nz = *reinterpret_cast<const f32x4 *>(sz + j0 + j + 4);
        float cm[4] = {3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};  // this lane's minimum per candidate
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float d[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float dx = qx[r] - cx[u], dy = qy[r] - cy[u], dz = qz[r] - cz[u];
            d[u] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            cm[u] = __builtin_fminf(cm[u], d[u]);
          }
          m[r] = __builtin_fminf(__builtin_fminf(m[r], d[0]), d[1]);
          m[r] = __builtin_fminf(__builtin_fminf(m[r], d[2]), d[3]);
        }
        // The fold runs on the bit patterns (non-negative floats order like unsigned integers): v_min_u32 takes a DPP
        // operand directly and needs no NaN canonicalisation, so every step below is one instruction.
        const unsigned c0 = __builtin_bit_cast(unsigned, cm[0]), c1 = __builtin_bit_cast(unsigned, cm[1]),
                       c2 = __builtin_bit_cast(unsigned, cm[2]), c3 = __builtin_bit_cast(unsigned, cm[3]);
        // quad butterfly: afterwards this lane holds the quad's minimum of candidate (lane & 3)
        unsigned k0 = b0 ? c1 : c0, k1 = b0 ? c3 : c2;
        const unsigned s0 = b0 ? c0 : c1, s1 = b0 ? c2 : c3;
        k0 = min(k0, dpp_u<0xB1>(s0));  // quad_perm [1,0,3,2]
        k1 = min(k1, dpp_u<0xB1>(s1));
        unsigned k = b1 ? k1 : k0;
        const unsigned sn = b1 ? k0 : k1;
        k = min(k, dpp_u<0x4E>(sn));   // quad_perm [2,3,0,1]
        k = min(k, dpp_u<0x124>(k));   // row_ror:4 (same quad position, next quad of the 16-lane row)
        k = min(k, dpp_u<0x128>(k));   // row_ror:8
        // across the four 16-lane rows: xor shuffles (v_permlane16/32_swap would avoid the LDS crossbar, but gave wrong
        // minima right behind the DPP steps — 75 % of the column results — and saved only 0.06 ms)
        k = min(k, (unsigned)__shfl_xor((int)k, 16, 64));
        k = min(k, (unsigned)__shfl_xor((int)k, 32, 64));
        // every lane learns the wave minimum of all four candidates; the first lane that holds it signs the entry
        const unsigned w0 = dpp_u<0x00>(k), w1 = dpp_u<0x55>(k), w2 = dpp_u<0xAA>(k), w3 = dpp_u<0xFF>(k);
        const unsigned l0 = __builtin_ctzll(__builtin_amdgcn_ballot_w64(c0 == w0));
        const unsigned l1 = __builtin_ctzll(__builtin_amdgcn_ballot_w64(c1 == w1));
        const unsigned l2 = __builtin_ctzll(__builtin_amdgcn_ballot_w64(c2 == w2));
        const unsigned l3 = __builtin_ctzll(__builtin_amdgcn_ballot_w64(c3 == w3));
        const unsigned ll = b1 ? (b0 ? l3 : l2) : (b0 ? l1 : l0);  // the winner of this lane's candidate (lane & 3)
        const unsigned long long packed = ((unsigned long long)k << 32) | (unsigned long long)(gwl | ll);
        if (lane < 4) atomicMin(&scol[j0 + j + lane], packed);
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
    __syncthreads();
    for (int i = threadIdx.x; i < ntile; i += 256) atomicMin(colmin + (long long)z * nq + t0 + i, scol[i]);
  }

  // Row direction: rescan the winning chunk (same arithmetic) for the first arg-min.
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int qi = q0 + r;
    if (qi >= np) continue;
    const int j0 = bchunk[r] * kNNChunk;
    float bd = 3.0e38f;
    int bj = j0;
    for (int j = 0; j < kNNChunk; ++j) {
      const int ja = j0 + j;
      const float cxv = ja < nq ? cb[ja * 3 + 0] : kFar, cyv = ja < nq ? cb[ja * 3 + 1] : kFar,
                  czv = ja < nq ? cb[ja * 3 + 2] : kFar;
      const float dx = qx[r] - cxv, dy = qy[r] - cyv, dz = qz[r] - czv;
      const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
      if (d < bd) {
        bd = d;
        bj = ja;
      }
    }
    dxy[(long long)z * np + qi] = bd;
    ixy[(long long)z * np + qi] = bj;
  }
}

// Column direction: (distance, wave, lane) -> index.  One thread per candidate.
template <int R>
__global__ __launch_bounds__(256) void nn2_cols_kernel(const float *__restrict__ xs, int np,
                                                       const float *__restrict__ ys, int nq, int batch, int nclouds,
                                                       const unsigned long long *__restrict__ colmin,
                                                       float *__restrict__ dyx, int32_t *__restrict__ iyx) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)nclouds * nq) return;
  const int z = (int)(t / nq), j = (int)(t - (long long)z * nq);
  const unsigned long long packed = colmin[t];
  const float dmin = __builtin_bit_cast(float, (unsigned)(packed >> 32));
  const unsigned id = (unsigned)packed;
  const int gw = id >> 6, ln = id & 63;
  const int base = (gw >> 2) * (256 * R) + (gw & 3) * (64 * R) + ln * R;
  const float *qb = xs + (long long)z * np * 3;
  const float *c = ys + ((long long)(z % batch) * nq + j) * 3;
  const float cxv = c[0], cyv = c[1], czv = c[2];
  int found = min(base, np - 1);
#pragma unroll
  for (int r = R - 1; r >= 0; --r) {
    const int qi = min(base + r, np - 1);
    const float dx = qb[qi * 3 + 0] - cxv, dy = qb[qi * 3 + 1] - cyv, dz = qb[qi * 3 + 2] - czv;
    const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    if (d == dmin) found = qi;  // descending r: the first query of the lane that reproduces the distance stays
  }
  dyx[t] = dmin;
  iyx[t] = found;
}

size_t chamfer_scratch_bytes(int draws, int batch, int q) {
  return (size_t)draws * batch * q * sizeof(unsigned long long);
}

template <int R>
static void launch_nn2_r(const float *x, int p, const float *y, int q, int batch, int nclouds, float *dxy, int32_t *ixy,
                         float *dyx, int32_t *iyx, unsigned long long *colmin, hipStream_t s) {
  dim3 grid(cdiv(p, 256 * R), nclouds);
  A3VT_LAUNCH((nn2_kernel<R>), grid, dim3(256), 0, s, x, p, y, q, batch, dxy, ixy, colmin);
  const long long total = (long long)nclouds * q;
  A3VT_LAUNCH((nn2_cols_kernel<R>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, x, p, y, q, batch, nclouds,
              colmin, dyx, iyx);
}

static int launch_nn2(const float *x, int p, const float *y, int q, int batch, int nclouds, float *dxy, int32_t *ixy,
                      float *dyx, int32_t *iyx, void *scratch, hipStream_t s) {
  auto *colmin = static_cast<unsigned long long *>(scratch);
  (void)hipGetLastError();
  if (hipMemsetAsync(colmin, 0xFF, (size_t)nclouds * q * sizeof(unsigned long long), s) != hipSuccess) {
    set_error("chamfer_fwd: clearing the column scratch failed");
    return -2;
  }
#ifdef A3VT_DBG_ENV   // variant builds only (tools/build_variants.sh env): the shipped library reads no environment variable
  static const int env_r = getenv("A3VT_NN_R") ? atoi(getenv("A3VT_NN_R")) : 0;  // developer override
#else
  constexpr int env_r = 0;
#endif
  int best_r = 4;
  long long best_cost = -1;
  // balance rule of launch_nn, plus the column fold: ~28 wave-instructions per 4 candidates whatever R is, i.e. about
  // 1.1 queries' worth — larger R amortises it (10,000 x 192: R = 10 -> 768 workgroups = 3 per CU, 3.9 ms; R = 5 4.1 ms)
  for (int r : {10, 8, 6, 5, 4, 3}) {
    const long long wgs = (long long)cdiv(p, 256 * r) * nclouds;
    const long long cost = ((wgs + 255) / 256) * (10 * r + 11);
    if (best_cost < 0 || cost < best_cost) best_cost = cost, best_r = r;
  }
  if (env_r) best_r = env_r;
  switch (best_r) {
    case 10: launch_nn2_r<10>(x, p, y, q, batch, nclouds, dxy, ixy, dyx, iyx, colmin, s); break;
    case 8: launch_nn2_r<8>(x, p, y, q, batch, nclouds, dxy, ixy, dyx, iyx, colmin, s); break;
    case 6: launch_nn2_r<6>(x, p, y, q, batch, nclouds, dxy, ixy, dyx, iyx, colmin, s); break;
    case 5: launch_nn2_r<5>(x, p, y, q, batch, nclouds, dxy, ixy, dyx, iyx, colmin, s); break;
    case 3: launch_nn2_r<3>(x, p, y, q, batch, nclouds, dxy, ixy, dyx, iyx, colmin, s); break;
    default: launch_nn2_r<4>(x, p, y, q, batch, nclouds, dxy, ixy, dyx, iyx, colmin, s); break;
  }
  A3VT_CHECK_LAUNCH();
  return 0;
}

// cd[b] = (1/draws) * sum_r ( mean_i dxy[r][b][i] + mean_j dyx[r][b][j] ) ; one workgroup per b, fixed order.
__global__ __launch_bounds__(1024) void chamfer_reduce_kernel(const float *__restrict__ dxy,
                                                              const float *__restrict__ dyx, int draws, int batch,
                                                              int p, int q, float *__restrict__ cd) {
  // 1024 threads per sample (only `batch` workgroups exist: the reduce is latency-bound, so each gets many loads in flight)
  __shared__ float red[16];
  const int b = blockIdx.x;
  float total = 0.f;
  for (int r = 0; r < draws; ++r) {
    float s1 = 0.f, s2 = 0.f;
    const float *a = dxy + ((long long)r * batch + b) * p;
    const float *bb = dyx + ((long long)r * batch + b) * q;
    for (int i = threadIdx.x; i < p; i += 1024) s1 += a[i];
    for (int i = threadIdx.x; i < q; i += 1024) s2 += bb[i];
    float v = s1 / (float)p + s2 / (float)q;
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];  // fixed order
    total += t;
  }
  if (threadIdx.x == 0) cd[b] = total / (float)draws;
}

size_t chamfer_workspace_bytes(int draws, int batch, int p, int q) {
  const size_t a = chamfer_scratch_bytes(draws, batch, q), b = nn_pruned_workspace_bytes(draws, batch, p, q);
  return a > b ? a : b;
}

int launch_chamfer_fwd(const float *x, const float *y, int draws, int batch, int p, int q, float *dxy, int32_t *ixy,
                       float *dyx, int32_t *iyx, float *cd, void *scratch, size_t scratch_bytes, int algo, hipStream_t s,
                       int y_batch) {
  if (p <= 0 || q <= 0 || draws <= 0 || batch <= 0) {
    set_error("chamfer_fwd: empty input (draws=%d batch=%d p=%d q=%d)", draws, batch, p, q);
    return -1;
  }
  // y may hold fewer clouds than x has meshes: mesh b is compared with y[b % y_batch] (the candidate-major batches of the
  // scoring loop: K candidates x E elements share the E ground-truth clouds — they are sorted / boxed once, not K times)
  if (y_batch <= 0) y_batch = batch;
  if (batch % y_batch != 0) {
    set_error("chamfer_fwd: batch=%d is not a multiple of y_batch=%d", batch, y_batch);
    return -1;
  }
  if (algo < NN_AUTO || algo > NN_PRUNED) {
    set_error("chamfer_fwd: unknown search algorithm %d", algo);
    return -1;
  }
  const bool fits_pruned = scratch && scratch_bytes >= nn_pruned_workspace_bytes(draws, batch, p, q);
  const bool fits_sweep = scratch && scratch_bytes >= chamfer_scratch_bytes(draws, batch, q) &&
                          (long long)cdiv(p, 256 * 3) * 4 < (1 << 26);
  if ((algo == NN_PRUNED && !fits_pruned) || (algo == NN_BRUTE_SWEEP && !fits_sweep)) {
    set_error("chamfer_fwd: workspace of %zu bytes is too small for search algorithm %d", scratch_bytes, algo);
    return -1;
  }
  if (algo == NN_AUTO) {
    // the sort and the per-wave block scan pay for themselves from a couple of thousand points per cloud on
#ifdef A3VT_DBG_ENV
    static const int env = getenv("A3VT_NN_ALGO") ? atoi(getenv("A3VT_NN_ALGO")) : 0;   // developer override
#else
    constexpr int env = 0;
#endif
    if (env == NN_BRUTE_TWO_PASS || (env == NN_BRUTE_SWEEP && fits_sweep) || (env == NN_PRUNED && fits_pruned)) algo = env;
    else algo = (fits_pruned && p >= 2048 && q >= 2048) ? NN_PRUNED : fits_sweep ? NN_BRUTE_SWEEP : NN_BRUTE_TWO_PASS;
  }
  if (algo == NN_PRUNED) {
    if (int rc = launch_nn_pruned(x, y, draws, batch, p, q, dxy, ixy, dyx, iyx, scratch, s, y_batch)) return rc;
  } else if (algo == NN_BRUTE_SWEEP) {
    // one pass over the distance matrix for both directions (rows = predicted clouds, columns = ground truth)
    if (int rc = launch_nn2(x, p, y, q, y_batch, draws * batch, dxy, ixy, dyx, iyx, scratch, s)) return rc;
  } else {
    // x -> y: queries = x clouds (draws*batch distinct), candidates = y[b]
    if (int rc = launch_nn(x, p, draws * batch, y, q, y_batch, draws * batch, dxy, ixy, s)) return rc;
    // y -> x: queries = y[b], candidates = x[r][b]
    if (int rc = launch_nn(y, q, y_batch, x, p, draws * batch, draws * batch, dyx, iyx, s)) return rc;
  }
  A3VT_LAUNCH(chamfer_reduce_kernel, dim3(batch), dim3(1024), 0, s, dxy, dyx, draws, batch, p, q, cd);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// Gradient.  With g = grad_cd[b] / draws:
//   gx[r][b][i] = g (2/P) (x_i - y_nn(i))  +  g (2/Q) sum_{j : nn(j) = i} (x_i - y_j)
//   gy[b][j]    = - sum_r [ g (2/Q) (x_nn_r(j) - y_j)  +  g (2/P) sum_{i : nn_r(i) = j} (x_i - y_j) ]
// Both are "a direct term per target point plus a scatter of the other cloud's nearest-neighbour pairs into it".  One
// kernel serves both: a workgroup owns a tile of ONE target cloud, accumulates the scattered differences of every source
// cloud in 64-bit fixed point in LDS (common.h: integer sums do not depend on the arrival order, so the gradient is
// reproducible bit for bit, unlike the float atomics this replaces), and writes each gradient element exactly once —
// direct term included, so nothing has to be zeroed or added to afterwards.  Any cloud size: the target is cut into
// tiles of <= kBwdTile points (24 bytes of LDS each); every tile's workgroup scans all source points and keeps those
// whose neighbour falls into its tile.
constexpr int kBwdTile = 6144;  // 144 KiB of accumulators

// grid = (target clouds, tiles).  Target cloud t: points tgt + t*nt*3; its sources: src + (s * src_stride_s +
// (t % src_mod)) * ns * 3 for s < nsrc.  idx_t2s[(s*idx_stride + t) * nt + i] = neighbour of target point i in source s,
// idx_s2t[(s*idx_stride + t) * ns + j] = neighbour of source point j (of source s) in the target cloud.
// out[t][i] = sign * sum_s ( ct * (T_i - S_nn(i)) + cs * sum_{j: nn(j)=i} (T_i - S_j) ),  ct = g*2/nt, cs = g*2/ns.
__global__ __launch_bounds__(1024) void chamfer_bwd_kernel(const float *__restrict__ tgt, int nt,
                                                           const float *__restrict__ src, int ns, int nsrc,
                                                           long long src_stride_s, int src_mod, long long idx_stride,
                                                           const int32_t *__restrict__ idx_t2s,
                                                           const int32_t *__restrict__ idx_s2t,
                                                           const float *__restrict__ gcd, int batch, float inv_draws,
                                                           float *__restrict__ out) {
  extern __shared__ long long facc[];  // [tile][3]
  __shared__ float red[2][16];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int tile = (nt + gridDim.y - 1) / gridDim.y;
  const int i0 = blockIdx.y * tile, i1 = min(nt, i0 + tile);
  const float *T = tgt + (long long)t * nt * 3;
  const float g = gcd[t % batch] * inv_draws;
  const float ct = g * (2.0f / (float)nt), cs = g * (2.0f / (float)ns);

  // bound on |T_i - S_j| per coordinate: max|T| + max|S| over the clouds involved (the same value in every tile)
  float mx = 0.f;
  for (int i = tid; i < nt * 3; i += 1024) mx = fmaxf(mx, fabsf(T[i]));
  float ms = 0.f;
  for (int s = 0; s < nsrc; ++s) {
    const float *S = src + (s * src_stride_s + (t % src_mod)) * ns * 3;
    for (int j = tid; j < ns * 3; j += 1024) ms = fmaxf(ms, fabsf(S[j]));
  }
  mx = wave_max(mx);
  ms = wave_max(ms);
  if ((tid & 63) == 0) red[0][tid >> 6] = mx, red[1][tid >> 6] = ms;
  for (int i = tid; i < (i1 - i0) * 3; i += 1024) facc[i] = 0;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 16; ++w) mx = fmaxf(mx, red[0][w]), ms = fmaxf(ms, red[1][w]);
  const float bound = mx + ms;
  const bool finite = bound < 3.0e38f;  // false for Inf / NaN coordinates: the gradient is then NaN, loudly
  const FixScale fs = fix_scale(finite ? bound : 0.f, (long long)nsrc * ns);

  for (int s = 0; s < nsrc; ++s) {
    const float *S = src + (s * src_stride_s + (t % src_mod)) * ns * 3;
    const int32_t *nn = idx_s2t + (s * idx_stride + t) * ns;
    for (int j = tid; j < ns; j += 1024) {
      const int i = nn[j];
      if (i < i0 || i >= i1) continue;
#pragma unroll
      for (int d = 0; d < 3; ++d) fix_add(facc + (i - i0) * 3 + d, fix_from(T[i * 3 + d] - S[j * 3 + d], fs));
    }
  }
  __syncthreads();
  float *o = out + (long long)t * nt * 3;
  for (int k = tid; k < (i1 - i0) * 3; k += 1024) {
    const int i = i0 + k / 3, d = k - (k / 3) * 3;
    float direct = 0.f;
    for (int s = 0; s < nsrc; ++s) {  // fixed order over the draws
      const float *S = src + (s * src_stride_s + (t % src_mod)) * ns * 3;
      const int j = idx_t2s[(s * idx_stride + t) * nt + i];
      direct += T[i * 3 + d] - S[j * 3 + d];
    }
    const float v = ct * direct + cs * fix_to(facc[k], fs);
    o[i0 * 3 + k] = finite ? v : __builtin_nanf("");
  }
}

int launch_chamfer_bwd(const float *x, const float *y, int draws, int batch, int p, int q, const int32_t *ixy,
                       const int32_t *iyx, const float *gcd, float *gx, float *gy, hipStream_t s) {
  static OncePerDevice once;
  once.run([] {
    (void)hipFuncSetAttribute((const void *)chamfer_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              kBwdTile * 3 * sizeof(long long));
  });
  const float inv_draws = 1.0f / (float)draws;
  {  // grad_x: targets = the draws*batch predicted clouds, one source each (y[b])
    const int tiles = cdiv(p, kBwdTile), tile = cdiv(p, tiles);
    A3VT_LAUNCH(chamfer_bwd_kernel, dim3(draws * batch, tiles), dim3(1024), (size_t)tile * 3 * sizeof(long long), s, x, p, y,
                q, 1, 0ll, batch, 0ll, ixy, iyx, gcd, batch, inv_draws, gx);
    A3VT_CHECK_LAUNCH();
  }
  if (gy) {  // grad_y: targets = the batch ground-truth clouds, sources = their `draws` predicted clouds
    const int tiles = cdiv(q, kBwdTile), tile = cdiv(q, tiles);
    // d cd / d y_j has the opposite sign of the pair differences (y_j - x_i) handled as (T - S) with T = y: the kernel
    // computes sum (y - x) terms, which IS the gradient w.r.t. y — no sign flip needed.
    A3VT_LAUNCH(chamfer_bwd_kernel, dim3(batch, tiles), dim3(1024), (size_t)tile * 3 * sizeof(long long), s, y, q, x, p,
                draws, (long long)batch, batch, (long long)batch, iyx, ixy, gcd, batch, inv_draws, gy);
    A3VT_CHECK_LAUNCH();
  }
  return 0;
}

}  // namespace a3vt
