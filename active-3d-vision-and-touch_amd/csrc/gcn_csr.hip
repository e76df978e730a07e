// gcn_csr.hip — CSR neighbour aggregation and the 3-channel output layer of the GCN stack (gfx950).
//
// Replaces the DENSE (N,N) adjacency products of reconstruction/vision/model.py:356,360
// (torch.matmul(adj, features[:, :, :length]) / torch.matmul(adj, features)) with gathers over the
// row-normalised CSR adjacency (nnz 17,922 on the 2562-vertex icosphere vs 6.6 M dense entries), fused
// with the partial bias add and ReLU of model.py:357-358,361,363.  All of these are HBM/L2-bound
// gathers of 400-byte rows; the neighbour rows of one mesh (<= 1 MB) live in the XCD's L2.
#include "common.h"
#include "kernels.h"

namespace a3vt {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// XCD-aware work mapping.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 names the XCD group,
// MI355X_MICROARCH.md "Workgroup dispatch"), each XCD has a private 4 MiB L2, and a neighbour gather only
// ever touches rows of its own mesh (<= 1 MB per mesh at 100 channels).  So mesh b is processed entirely by
// the workgroups of XCD group b % 8: the mesh's rows are pulled into ONE L2 and every re-read (7x on the
// icosphere) hits there, instead of every L2 thrashing over every mesh and falling through to the Infinity
// Cache.  Placement only affects speed, never correctness.
struct XcdWalk {
  int xcd, nloc, stride;       // this workgroup's XCD group, its index inside the group, workgroups per group
  long long ngroups;           // 8-vertex groups owned by the XCD group
  int gps, nmesh;              // groups per mesh, meshes owned
  __device__ XcdWalk(int batch, int n_vert, int skip = 0) {   // skip: workgroups in front that do other work (% 8 == 0)
    const int bid = blockIdx.x - skip, grid = gridDim.x - skip;
    xcd = bid & 7;
    nloc = bid >> 3;
    stride = (grid + 7 - xcd) >> 3;  // workgroups with this blockIdx % 8
    gps = (n_vert + 7) >> 3;
    nmesh = batch > xcd ? (batch - xcd + 7) >> 3 : 0;
    ngroups = (long long)nmesh * gps;
  }
  // vertex handled by sub-group `sub` (0..7) of group g; returns false past the mesh end
  __device__ bool locate(long long g, int sub, int n_vert, long long &b, int &v) const {
    b = xcd + 8 * (g / gps);
    v = (int)(g % gps) * 8 + sub;
    return v < n_vert;
  }
};

// Weighted sum of neighbour rows for one vertex, computed by a half-wave (32 lanes): the row's column
// indices and weights are fetched 32 at a time with ONE coalesced load each and handed around with
// shuffles, so the only dependent memory round trip per neighbour is the 16-byte row gather itself.
__device__ __forceinline__ f32x4 gather_row(const float *__restrict__ base, long long ld, int ch, bool lane_on,
                                            int e0, int e1, int hl, const int32_t *__restrict__ colidx,
                                            const float *__restrict__ val) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int eb = e0; eb < e1; eb += 32) {
    const int n = min(32, e1 - eb);
    const int myc = hl < n ? colidx[eb + hl] : 0;
    const float myw = hl < n ? val[eb + hl] : 0.f;
    int j = 0;
    for (; j + 3 < n; j += 4) {  // 4 neighbour rows in flight (8 costs occupancy and loses)
      const int c0 = __shfl(myc, j, 32), c1 = __shfl(myc, j + 1, 32), c2 = __shfl(myc, j + 2, 32),
                c3 = __shfl(myc, j + 3, 32);
      const float w0 = __shfl(myw, j, 32), w1 = __shfl(myw, j + 1, 32), w2 = __shfl(myw, j + 2, 32),
                  w3 = __shfl(myw, j + 3, 32);
      if (lane_on) {
        const f32x4 r0 = *reinterpret_cast<const f32x4 *>(base + c0 * ld + ch);
        const f32x4 r1 = *reinterpret_cast<const f32x4 *>(base + c1 * ld + ch);
        const f32x4 r2 = *reinterpret_cast<const f32x4 *>(base + c2 * ld + ch);
        const f32x4 r3 = *reinterpret_cast<const f32x4 *>(base + c3 * ld + ch);
        acc += w0 * r0;
        acc += w1 * r1;
        acc += w2 * r2;
        acc += w3 * r3;
      }
    }
    if (j < n) {  // one to three rows left: requested together as well (one at a time, a 6- or 7-edge row — every row of
      // an icosphere — paid two or three more dependent round trips); same order of accumulation
      const bool p1 = j + 1 < n, p2 = j + 2 < n;
      const int c0 = __shfl(myc, j, 32), c1 = __shfl(myc, j + 1, 32), c2 = __shfl(myc, j + 2, 32);
      const float w0 = __shfl(myw, j, 32), w1 = __shfl(myw, j + 1, 32), w2 = __shfl(myw, j + 2, 32);
      if (lane_on) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 r0 = *reinterpret_cast<const f32x4 *>(base + c0 * ld + ch);
        const f32x4 r1 = p1 ? *reinterpret_cast<const f32x4 *>(base + c1 * ld + ch) : z;
        const f32x4 r2 = p2 ? *reinterpret_cast<const f32x4 *>(base + c2 * ld + ch) : z;
        acc += w0 * r0;
        if (p1) acc += w1 * r1;
        if (p2) acc += w2 * r2;
      }
    }
  }
  return acc;
}

// ------------------------------------------------------------------------------------------------
// Hub rows.  The fused vision + touch graphs link every seam vertex to the centre of every touch chart
// (utility/utils.py:119-128), so a handful of rows have ~1150 neighbours while the rest have 5-25.  One half-wave
// walking such a row serially takes ~290 us and sets the kernel time (5x the uniform-graph time).  Rows with more than
// kHeavyDeg neighbours are therefore listed once per call (csr_heavy_list_kernel) and handled by a second launch in
// which a whole workgroup shares one (mesh, row): its 8 half-waves take an eighth of the edge list each and the
// partial sums are combined through LDS in a fixed order.
// ------------------------------------------------------------------------------------------------
constexpr int kHeavyDeg = 64;

// One workgroup scans all rows (n_vert is a few thousand) and compacts the hub rows through an LDS counter: no
// memset, no global atomics, nothing a HIP-graph capture could object to.  heavy[0] = count, heavy[64..] = rows.
__global__ __launch_bounds__(1024) void csr_heavy_list_kernel(const int32_t *__restrict__ rowptr, int n_vert,
                                                              int32_t *__restrict__ heavy) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  for (int v = threadIdx.x; v < n_vert; v += 1024)
    if (rowptr[v + 1] - rowptr[v] > kHeavyDeg) heavy[64 + atomicAdd(&cnt, 1)] = v;
  __syncthreads();
  if (threadIdx.x == 0) heavy[0] = cnt;
}

size_t csr_heavy_scratch_ints(int n_vert) { return (size_t)n_vert + 64; }
int csr_heavy_degree() { return kHeavyDeg; }

int launch_csr_heavy_list(const int32_t *rowptr, int n_vert, int32_t *heavy, hipStream_t s) {
  A3VT_LAUNCH(csr_heavy_list_kernel, dim3(1), dim3(1024), 0, s, rowptr, n_vert, heavy);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// mode 0: forward epilogue (bias, optional ReLU, sign bytes).  mode 1: backward (A^T gather; channels [c, cpad) pass
// the row's own gradient through).  Runs in the FIRST kCsrHubWgs workgroups of the aggregation launch (they start first and
// finish under the row walk of the others; as a launch of its own the hub pass cost 16-18 us per layer and direction).
constexpr int kCsrHubWgs = 256;   // multiple of 8: the row walk's blockIdx -> XCD mapping is unchanged behind them

template <int MODE>
__device__ __forceinline__ void csr_hub_rows(f32x4 (*red)[32], const float *__restrict__ src, int ld_src,
                                             const float *__restrict__ bias, int c, int cpad,
                                             const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                             const float *__restrict__ val, int n_vert, int batch,
                                             const int32_t *__restrict__ heavy, float *__restrict__ dst, int ld_dst,
                                             uint8_t *__restrict__ maskb, int mld, int relu) {
  const int hl = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const int count = heavy[0];
  const int width = MODE == 0 ? c : cpad;
  for (long long item = blockIdx.x; item < (long long)count * batch; item += kCsrHubWgs) {
    const int v = heavy[64 + (int)(item % count)];
    const long long b = item / count, row = b * n_vert + v;
    const float *sb = src + b * n_vert * (long long)ld_src;
    const int e0 = rowptr[v], e1 = rowptr[v + 1];
    const int per = ((e1 - e0 + 7) / 8 + 3) & ~3;  // edges per half-wave, a multiple of the gather's unroll
    const int s0 = min(e1, e0 + sub * per), s1 = min(e1, s0 + per);
    for (int ch0 = 0; ch0 < width; ch0 += 128) {
      const int ch = ch0 + hl * 4;
      const bool on = ch < width;
      red[sub][hl] = gather_row(sb, ld_src, ch, on, s0, s1, hl, colidx, val);
      __syncthreads();
      if (sub == 0 && on) {
        f32x4 acc = red[0][hl];
#pragma unroll 2
        for (int r = 1; r < 8; ++r) acc += red[r][hl];
        float *o = dst + row * ld_dst + ch;
        if (MODE == 0) {
          unsigned bits = 0;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float pre = ch + t < c ? acc[t] + bias[ch + t] : 0.f;
            const float out = (pre > 0.f || !relu) ? pre : 0.f;
            bits |= (pre > 0.f ? 1u : 0u) << t;
            if (ch + t < c) o[t] = out;
          }
          if (maskb) maskb[row * mld + (ch >> 2)] = (uint8_t)bits;
        } else {
          const f32x4 own = *reinterpret_cast<const f32x4 *>(sb + (long long)v * ld_src + ch);
#pragma unroll
          for (int t = 0; t < 4; ++t) o[t] = ch + t < c ? acc[t] : own[t];
        }
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Forward: Y[m][ch] = relu( sum_e val[e] * Za[b][col[e]][ch] + bias[ch] ), ch < c.
// A half-wave (32 lanes) owns one vertex; lane l handles channels 4l..4l+3 (16-byte loads of the
// neighbour row), so up to 128 aggregated channels are covered per pass.
// ------------------------------------------------------------------------------------------------
template <bool HUB>   // HUB: the first kCsrHubWgs workgroups take the hub rows (own instantiation: graphs without hubs keep
                      // the smaller register allocation)
__global__ __launch_bounds__(256) void csr_fwd_kernel(const float *__restrict__ za, int ldza,
                                                      const float *__restrict__ bias, int c,
                                                      const int32_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ colidx,
                                                      const float *__restrict__ val, int n_vert, long long m,
                                                      float *__restrict__ y, int ldy,
                                                      uint8_t *__restrict__ maskb, int mld, int relu,
                                                      int heavy_thresh, const int32_t *__restrict__ heavy) {
  constexpr int hub = HUB ? kCsrHubWgs : 0;
  if (HUB && (int)blockIdx.x < hub) {
    __shared__ f32x4 red[8][32];
    csr_hub_rows<0>(red, za, ldza, bias, c, (c + 3) & ~3, rowptr, colidx, val, n_vert, (int)(m / n_vert), heavy, y, ldy, maskb,
                    mld, relu);
    return;
  }
  const int hl = threadIdx.x & 31;
  const XcdWalk w((int)(m / n_vert), n_vert, hub);
  float bsv[4] = {0.f, 0.f, 0.f, 0.f};  // bias of this lane's 4 channels (c <= 128: one pass)
  if (c <= 128) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (hl * 4 + t < c) bsv[t] = bias[hl * 4 + t];
  }
  for (long long g = w.nloc; g < w.ngroups; g += w.stride) {
    long long b;
    int v;
    if (!w.locate(g, threadIdx.x >> 5, n_vert, b, v)) continue;  // uniform per half-wave
    const long long row = b * n_vert + v;
    const float *zb = za + b * n_vert * (long long)ldza;
    const int e0 = rowptr[v], e1 = rowptr[v + 1];
    if (e1 - e0 > heavy_thresh) continue;  // hub row: csr_hub_rows spreads it over a whole workgroup
    for (int ch0 = 0; ch0 < c; ch0 += 128) {
      const int ch = ch0 + hl * 4;
      const bool on = ch < c;
      const f32x4 acc = gather_row(zb, ldza, ch, on, e0, e1, hl, colidx, val);
      if (!on) continue;
#ifdef A3VT_DBG_CSR_NOSTORE   // timing-only ablation (tools/build_variants.sh csr): every gathered value stays live, nothing is stored
      if (acc[0] + acc[1] + acc[2] + acc[3] != 1.2345e-33f) continue;
#endif
      float *yo = y + row * ldy + ch;
      unsigned bits = 0;
      f32x4 o;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float pre = ch + t < c ? acc[t] + (c <= 128 ? bsv[t] : bias[ch + t]) : 0.f;
        o[t] = (pre > 0.f || !relu) ? pre : 0.f;
        bits |= (pre > 0.f ? 1u : 0u) << t;
      }
      if (ch + 3 < c && (ldy & 3) == 0) {
        *reinterpret_cast<f32x4 *>(yo) = o;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (ch + t < c) yo[t] = o[t];
      }
      if (maskb) maskb[row * mld + (ch >> 2)] = (uint8_t)bits;  // ReLU sign of the aggregated channels
    }
  }
}

int launch_csr_fwd(const float *za, int ldza, const float *bias, int c, const int32_t *rowptr, const int32_t *col,
                   const float *val, const int32_t *heavy, int n_vert, int batch, float *y, int ldy, uint8_t *maskb,
                   int mld, int relu, hipStream_t s) {
  if (ldza % 4 != 0 || ldza < pad4(c)) {
    set_error("csr_fwd: ldza=%d must be a multiple of 4 and >= pad4(c=%d)", ldza, c);
    return -1;
  }
  const long long m = (long long)batch * n_vert;
  const int grid = (int)(cdiv(m, 8) < 4096 ? (cdiv(m, 8) + 7) / 8 * 8 : 4096);  // multiple of 8: whole XCD groups
  if (heavy)
    A3VT_LAUNCH(csr_fwd_kernel<true>, dim3(grid + kCsrHubWgs), dim3(256), 0, s, za, ldza, bias, c, rowptr, col, val, n_vert, m,
                y, ldy, maskb, mld, relu, kHeavyDeg, heavy);
  else
    A3VT_LAUNCH(csr_fwd_kernel<false>, dim3(grid), dim3(256), 0, s, za, ldza, bias, c, rowptr, col, val, n_vert, m, y, ldy,
                maskb, mld, relu, 0x7fffffff, heavy);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Backward: dZa[m][ch] = sum_e valT[e] * G[b][colT[e]][ch]   (ch < c;  A^T gather)
//           dZa[m][ch] = G[m][ch]                              (c <= ch < cpad: pass-through columns)
//           db_slab[block][ch] = sum over the block's rows of G[m][ch]   (ch < c)
// Persistent blocks (grid-stride over groups of 8 rows) so the bias-gradient partials stay few.
// ------------------------------------------------------------------------------------------------
constexpr int kCsrBwdMaxBlocks = 2048;

template <bool HUB>
__global__ __launch_bounds__(256) void csr_bwd_kernel(const float *__restrict__ g, int ldg, int c, int cpad,
                                                      const int32_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ colidx,
                                                      const float *__restrict__ val, int n_vert, long long m,
                                                      float *__restrict__ dza, int lddza,
                                                      float *__restrict__ db_slab, int heavy_thresh,
                                                      const int32_t *__restrict__ heavy) {
  __shared__ float red[8][128];
  constexpr int hub = HUB ? kCsrHubWgs : 0;
  if (HUB && (int)blockIdx.x < hub) {
    static_assert(sizeof(f32x4) * 8 * 32 <= sizeof(red), "hub rows reuse the bias-gradient staging");
    csr_hub_rows<1>(reinterpret_cast<f32x4(*)[32]>(&red[0][0]), g, ldg, nullptr, c, cpad, rowptr, colidx, val, n_vert,
                    (int)(m / n_vert), heavy, dza, lddza, nullptr, 0, 0);
    return;
  }
  const int hl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  for (int ch0 = 0; ch0 < cpad; ch0 += 128) {
    const int ch = ch0 + hl * 4;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const bool on = ch < cpad;  // all 32 lanes of the half-wave stay in the loop: gather_row shuffles across them
    const XcdWalk w((int)(m / n_vert), n_vert, hub);
    for (long long gi = w.nloc; gi < w.ngroups; gi += w.stride) {
      long long b;
      int v;
      if (!w.locate(gi, grp, n_vert, b, v)) continue;  // uniform per half-wave
      const long long row = b * n_vert + v;
      const float *gb = g + b * n_vert * (long long)ldg;
      f32x4 own = {0.f, 0.f, 0.f, 0.f};
      if (on) own = *reinterpret_cast<const f32x4 *>(gb + (long long)v * ldg + ch);
      bsum += own;
      const int e0 = rowptr[v], e1 = rowptr[v + 1];
      if (e1 - e0 > heavy_thresh) continue;  // hub row: gathered and stored by csr_hub_rows (its bias share is counted above)
      const f32x4 acc = gather_row(gb, ldg, ch, on, e0, e1, hl, colidx, val);
      if (on) {
        f32x4 out;
#pragma unroll
        for (int t = 0; t < 4; ++t) out[t] = ch + t < c ? acc[t] : own[t];
        *reinterpret_cast<f32x4 *>(dza + row * lddza + ch) = out;
      }
    }
    // block partial of the bias gradient
#pragma unroll
    for (int t = 0; t < 4; ++t) red[grp][hl * 4 + t] = bsum[t];
    __syncthreads();
    if (threadIdx.x < 128 && ch0 + threadIdx.x < cpad) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += red[r][threadIdx.x];
      db_slab[(size_t)(blockIdx.x - hub) * cpad + ch0 + threadIdx.x] = s;
    }
    __syncthreads();
  }
}

int csr_bwd_num_slabs(int batch, int n_vert) {
  const long long groups = ((long long)batch * n_vert + 7) / 8;
  return (int)(groups < kCsrBwdMaxBlocks ? (groups + 7) / 8 * 8 : kCsrBwdMaxBlocks);  // multiple of 8 (XCD groups)
}

int launch_csr_bwd(const float *g, int ldg, int c, const int32_t *rowptrT, const int32_t *colT, const float *valT,
                   const int32_t *heavyT, int n_vert, int batch, float *dza, int lddza, float *db_slab, hipStream_t s) {
  const int cpad = pad4(c);
  if (ldg % 4 != 0 || lddza % 4 != 0 || lddza < cpad || ldg < cpad) {
    set_error("csr_bwd: ldg=%d lddza=%d c=%d violate alignment rules", ldg, lddza, c);
    return -1;
  }
  const long long m = (long long)batch * n_vert;
  const int slabs = csr_bwd_num_slabs(batch, n_vert);
  if (heavyT)
    A3VT_LAUNCH(csr_bwd_kernel<true>, dim3(slabs + kCsrHubWgs), dim3(256), 0, s, g, ldg, c, cpad, rowptrT, colT, valT, n_vert,
                m, dza, lddza, db_slab, kHeavyDeg, heavyT);
  else
    A3VT_LAUNCH(csr_bwd_kernel<false>, dim3(slabs), dim3(256), 0, s, g, ldg, c, cpad, rowptrT, colT, valT, n_vert, m, dza,
                lddza, db_slab, 0x7fffffff, heavyT);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Output layer (out_features = 3): z3 = X W (thin product, HBM-bound on X), then the 3-channel
// aggregation + bias (no activation).  16 lanes share one row of X; each lane keeps the W rows of
// its k-pieces in registers.  k <= 320.
// ------------------------------------------------------------------------------------------------
constexpr int kThinPieces = 5;  // 16 lanes * 5 pieces * 4 floats = 320 channels max
// Both kernels are grid-stride loops with a per-block preamble (60 weight registers per lane): the grids are what is
// resident at once — 3 workgroups per CU for thin_bwd (155 VGPRs), 5 for thin_fwd (88) — so no block waits for a slot
// and then runs alone (thin_bwd: 1024 blocks at 768 resident took two rounds, 128 us per call; 768 take 99.  thin_fwd: 73 -> 52 us).
constexpr int kThinBlocks = 512;
constexpr int kThinFwdBlocks = 768;
constexpr int kThinFwdRows = 3;
int thin_num_slabs() { return kThinBlocks; }

// Hybrid rows (the stack's channel-sliced path): columns [0, 4 xq_quads) of X come from the quad-major array xq
// [m / n_vert][xq_quads][n_vert] float4 (the aggregation kernel's output), the rest from x (pre-offset, row stride ldx).
// (mesh, vertex) of a row that advances by a fixed step: one 64-bit division at the start, then adds and one compare per step
// (a 64-bit `row / n_vert` per row cost thin_bwd 10 us of its 167)
struct MeshWalk {
  int b, v, sb, sv, n;
  __device__ MeshWalk(long long row0, long long step, int n_vert) {
    n = n_vert > 0 ? n_vert : 1;
    b = (int)(row0 / n);
    v = (int)(row0 - (long long)b * n);
    sb = (int)(step / n);
    sv = (int)(step - (long long)sb * n);
  }
  __device__ void next() {
    b += sb;
    v += sv;
    if (v >= n) { v -= n; ++b; }
  }
};

__global__ __launch_bounds__(256) void thin_fwd_kernel(const float *__restrict__ x, int ldx, int k,
                                                       const float *__restrict__ w, long long m,
                                                       float *__restrict__ z3, const float *__restrict__ xq,
                                                       int xq_quads, int n_vert) {
  const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;  // 16 row groups per block
  float wr[kThinPieces][4][3];
#pragma unroll
  for (int p = 0; p < kThinPieces; ++p)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int kk = (p * 16 + l16) * 4 + t;
#pragma unroll
      for (int j = 0; j < 3; ++j) wr[p][t][j] = kk < k ? w[kk * 3 + j] : 0.f;
    }
  // kThinFwdRows rows per trip: that many times five 16-byte loads in flight per lane (one row per trip left the memory
  // path at 2.6 TB/s: 75 us per call; two 47)
  const long long step = (long long)gridDim.x * 16;
  MeshWalk mw((long long)blockIdx.x * 16 + grp, step, n_vert);
  for (long long row0 = (long long)blockIdx.x * 16 + grp; row0 < m; row0 += kThinFwdRows * step) {
    long long rows[kThinFwdRows];
    f32x4 xv[kThinFwdRows][kThinPieces];
#pragma unroll
    for (int u = 0; u < kThinFwdRows; ++u) {
      const long long row = row0 + u * step;
      rows[u] = row;
      const bool live = row < m;
      const float *xqr = xq ? xq + ((size_t)mw.b * xq_quads * n_vert + (size_t)mw.v) * 4 : nullptr;   // this row in quad plane 0
#pragma unroll
      for (int p = 0; p < kThinPieces; ++p) {
        const int kk = (p * 16 + l16) * 4;
        xv[u][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kk < k && live) {
          const float *src = (xq && kk < xq_quads * 4) ? xqr + (size_t)(kk >> 2) * n_vert * 4 : x + row * ldx + kk;
          xv[u][p] = *reinterpret_cast<const f32x4 *>(src);
        }
      }
      mw.next();
    }
#pragma unroll
    for (int u = 0; u < kThinFwdRows; ++u) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int p = 0; p < kThinPieces; ++p) {
        const int kk = (p * 16 + l16) * 4;
        if (kk < k) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            s0 += xv[u][p][t] * wr[p][t][0];
            s1 += xv[u][p][t] * wr[p][t][1];
            s2 += xv[u][p][t] * wr[p][t][2];
          }
        }
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off, 16);
        s1 += __shfl_xor(s1, off, 16);
        s2 += __shfl_xor(s2, off, 16);
      }
      if (l16 == 0 && rows[u] < m) *reinterpret_cast<f32x4 *>(z3 + rows[u] * 4) = f32x4{s0, s1, s2, 0.f};
    }
  }
}

// out[m][0..2] = sum_e val[e] * z[b][col[e]][0..2] (+ bias) ; z rows are float4 (4th lane unused)
__global__ __launch_bounds__(256) void csr3_kernel(const float *__restrict__ z, const float *__restrict__ bias,
                                                   const int32_t *__restrict__ rowptr,
                                                   const int32_t *__restrict__ colidx,
                                                   const float *__restrict__ val, int n_vert, long long m,
                                                   float *__restrict__ out, int ldo, int heavy_thresh) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= m) return;
  const long long b = row / n_vert;
  const int v = (int)(row - b * n_vert);
  const float *zb = z + b * n_vert * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (rowptr[v + 1] - rowptr[v] > heavy_thresh) return;  // hub row: csr3_heavy_kernel
  for (int e = rowptr[v]; e < rowptr[v + 1]; ++e)
    acc += val[e] * *reinterpret_cast<const f32x4 *>(zb + (long long)colidx[e] * 4);
  if (bias) {
    acc[0] += bias[0];
    acc[1] += bias[1];
    acc[2] += bias[2];
  }
  if (ldo == 4) {
    acc[3] = 0.f;
    *reinterpret_cast<f32x4 *>(out + row * 4) = acc;
  } else {
    out[row * ldo + 0] = acc[0];
    out[row * ldo + 1] = acc[1];
    out[row * ldo + 2] = acc[2];
  }
}

// Hub rows of the 3-channel aggregation: one wave per (mesh, row), lanes stride over the edge list.
__global__ __launch_bounds__(256) void csr3_heavy_kernel(const float *__restrict__ z, const float *__restrict__ bias,
                                                         const int32_t *__restrict__ rowptr,
                                                         const int32_t *__restrict__ colidx,
                                                         const float *__restrict__ val, int n_vert, int batch,
                                                         const int32_t *__restrict__ heavy, float *__restrict__ out,
                                                         int ldo) {
  const int lane = threadIdx.x & 63;
  const int count = heavy[0];
  for (long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); item < (long long)count * batch;
       item += (long long)gridDim.x * 4) {
    const int v = heavy[64 + (int)(item % count)];
    const long long b = item / count, row = b * n_vert + v;
    const float *zb = z + b * n_vert * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int e = rowptr[v] + lane; e < rowptr[v + 1]; e += 64)
      acc += val[e] * *reinterpret_cast<const f32x4 *>(zb + (long long)colidx[e] * 4);
    acc[0] = wave_sum(acc[0]);
    acc[1] = wave_sum(acc[1]);
    acc[2] = wave_sum(acc[2]);
    if (lane == 0) {
      if (bias) {
        acc[0] += bias[0];
        acc[1] += bias[1];
        acc[2] += bias[2];
      }
      out[row * ldo + 0] = acc[0];
      out[row * ldo + 1] = acc[1];
      out[row * ldo + 2] = acc[2];
      if (ldo == 4) out[row * 4 + 3] = 0.f;
    }
  }
}

// The same aggregation through the split D^-1 (P + J) of the fused vision + touch matrix (a3vt_adj_split; gcn_csrqs.hip has the
// algebra): a workgroup per (mesh, part of its vertices) keeps the mesh's [n_vert] float4 rows in LDS, forms the two class
// sums, and walks the short rows of P.  TRANSPOSED = the product with A^T: P and J are symmetric, so it is the same walk
// over rows scaled on their way in instead of sums scaled on their way out.  No hub list, no second launch.
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void csr3s_kernel(const float *__restrict__ z, const float *__restrict__ bias,
                                                    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                                    const float *__restrict__ scale, const uint8_t *__restrict__ cls,
                                                    int n_vert, int parts, float *__restrict__ out, int ldo) {
  extern __shared__ __attribute__((aligned(16))) float lds3[];
  __shared__ f32x4 red[2][4];
  f32x4 *zs = reinterpret_cast<f32x4 *>(lds3);
  const int b = blockIdx.x / parts, part = blockIdx.x - b * parts;
  const f32x4 *zb = reinterpret_cast<const f32x4 *>(z) + (size_t)b * n_vert;
  f32x4 ps = {0.f, 0.f, 0.f, 0.f}, pc = ps;
  for (int v = threadIdx.x; v < n_vert; v += 256) {
    f32x4 x = zb[v];
    if (TRANSPOSED) x *= scale[v];
    zs[v] = x;
    const int cl = cls[v];
    if (cl == 1) ps += x;
    if (cl == 2) pc += x;
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) ps[t] = wave_sum(ps[t]), pc[t] = wave_sum(pc[t]);
  if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = ps, red[1][threadIdx.x >> 6] = pc;
  __syncthreads();
  const f32x4 sig_s = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
  const f32x4 sig_c = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
  const int v0 = (int)((long long)part * n_vert / parts), v1 = (int)((long long)(part + 1) * n_vert / parts);
  for (int v = v0 + threadIdx.x; v < v1; v += 256) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) acc += zs[colidx[e]];
    const int cl = cls[v];
    if (cl == 1) acc += sig_c;
    if (cl == 2) acc += sig_s;
    if (!TRANSPOSED) acc *= scale[v];
    if (bias) {
      acc[0] += bias[0];
      acc[1] += bias[1];
      acc[2] += bias[2];
    }
    const long long row = (long long)b * n_vert + v;
    if (ldo == 4) {
      acc[3] = 0.f;
      *reinterpret_cast<f32x4 *>(out + row * 4) = acc;
    } else {
      out[row * ldo + 0] = acc[0];
      out[row * ldo + 1] = acc[1];
      out[row * ldo + 2] = acc[2];
    }
  }
}

int launch_csr3(const float *z, const float *bias, const int32_t *rowptr, const int32_t *col, const float *val,
                       const int32_t *heavy, int n_vert, int batch, float *out, int ldo, hipStream_t s, const SplitRef *sp,
                       bool transposed) {
  const long long m = (long long)batch * n_vert;
  if (sp && (size_t)n_vert * 16 <= 60 * 1024) {
    int parts = 512 / batch;   // ~two workgroups per CU
    parts = parts < 1 ? 1 : parts > 8 ? 8 : parts;
    if (transposed)
      A3VT_LAUNCH(csr3s_kernel<true>, dim3(batch * parts), dim3(256), (size_t)n_vert * 16, s, z, bias, sp->rowptr, sp->col,
                  sp->scale, sp->cls, n_vert, parts, out, ldo);
    else
      A3VT_LAUNCH(csr3s_kernel<false>, dim3(batch * parts), dim3(256), (size_t)n_vert * 16, s, z, bias, sp->rowptr, sp->col,
                  sp->scale, sp->cls, n_vert, parts, out, ldo);
    A3VT_CHECK_LAUNCH();
    return 0;
  }
  A3VT_LAUNCH(csr3_kernel, dim3(cdiv(m, 256)), dim3(256), 0, s, z, bias, rowptr, col, val, n_vert, m, out, ldo,
              heavy ? kHeavyDeg : 0x7fffffff);
  A3VT_CHECK_LAUNCH();
  if (heavy) {
    A3VT_LAUNCH(csr3_heavy_kernel, dim3(64), dim3(256), 0, s, z, bias, rowptr, col, val, n_vert, batch, heavy, out, ldo);
    A3VT_CHECK_LAUNCH();
  }
  return 0;
}

int launch_thin_fwd(const float *x, int ldx, int k, const float *w, const float *bias, const int32_t *rowptr,
                    const int32_t *col, const float *val, const int32_t *heavy, int n_vert, int batch, float *z3,
                    float *update, const float *xq, int xq_quads, hipStream_t s, const SplitRef *sp) {
  if (k > kThinPieces * 64 || ldx % 4 != 0) {
    set_error("thin_fwd: k=%d (max %d) ldx=%d unsupported", k, kThinPieces * 64, ldx);
    return -1;
  }
  const long long m = (long long)batch * n_vert;
  const int grid = (int)(cdiv(m, 16) < kThinFwdBlocks ? cdiv(m, 16) : kThinFwdBlocks);
  A3VT_LAUNCH(thin_fwd_kernel, dim3(grid), dim3(256), 0, s, x, ldx, k, w, m, z3, xq, xq_quads, n_vert);
  A3VT_CHECK_LAUNCH();
  return launch_csr3(z3, bias, rowptr, col, val, heavy, n_vert, batch, update, 3, s, sp, false);
}

// Backward of the output layer.  dz3 = A^T dU (csr3 with the transposed CSR, no bias), then one pass over X:
//   G_prev[m][k] = (sum_j dz3[m][j] W[k][j]) * (apply_mask ? X[m][k] > 0 : 1)
//   dW[k][j]    += X[m][k] dz3[m][j]      (register partials -> LDS -> slab per block)
//   db[j]        = sum_m dU[m][j]         (slab per block)
__global__ __launch_bounds__(256) void thin_bwd_kernel(const float *__restrict__ x, int ldx, int k,
                                                       const float *__restrict__ w,
                                                       const float *__restrict__ dz3,
                                                       const float *__restrict__ du, long long m, int apply_mask,
                                                       float *__restrict__ gprev, int ldg, int n_store,
                                                       float *__restrict__ dw_slab,
                                                       float *__restrict__ db_slab, float *__restrict__ gq, int nq,
                                                       int n_vert, const float *__restrict__ xq, int xq_quads) {
  __shared__ float red[16][kThinPieces * 64 * 3 / 16 + 1];  // [row group][this lane-column's 60 partials] per l16 pass
  const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  float wr[kThinPieces][4][3], dwp[kThinPieces][4][3];
#pragma unroll
  for (int p = 0; p < kThinPieces; ++p)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int kk = (p * 16 + l16) * 4 + t;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        wr[p][t][j] = kk < k ? w[kk * 3 + j] : 0.f;
        dwp[p][t][j] = 0.f;
      }
    }
  float db0 = 0.f, db1 = 0.f, db2 = 0.f;
  // two rows per trip (ten 16-byte loads of X in flight per lane), as in thin_fwd
  const long long step = (long long)gridDim.x * 16;
  MeshWalk mw((long long)blockIdx.x * 16 + grp, step, n_vert);
  for (long long row0 = (long long)blockIdx.x * 16 + grp; row0 < m; row0 += 2 * step) {
    f32x4 xv[2][kThinPieces], d[2];
    float *gr[2], *gqr[2];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long long row = row0 + u * step;
      live[u] = row < m;
      d[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (live[u]) {
        d[u] = *reinterpret_cast<const f32x4 *>(dz3 + row * 4);
        if (l16 == 0) {
          db0 += du[row * 3 + 0];
          db1 += du[row * 3 + 1];
          db2 += du[row * 3 + 2];
        }
      }
      gr[u] = gprev + row * ldg;
      // quad-major copy of the aggregated-channel columns for csrq_kernel<1> (launch_csrq_bwd), when asked for
      gqr[u] = gq ? gq + ((size_t)mw.b * nq * n_vert + (size_t)mw.v) * 4 : nullptr;
      const float *xqr = xq ? xq + ((size_t)mw.b * xq_quads * n_vert + (size_t)mw.v) * 4 : nullptr;
#pragma unroll
      for (int p = 0; p < kThinPieces; ++p) {
        const int kk = (p * 16 + l16) * 4;
        xv[u][p] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kk < k && live[u]) {
          const float *src = (xq && kk < xq_quads * 4) ? xqr + (size_t)(kk >> 2) * n_vert * 4 : x + row * ldx + kk;
          xv[u][p] = *reinterpret_cast<const f32x4 *>(src);
        }
      }
      mw.next();
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!live[u]) continue;   // (uniform per 16-lane group; a dead row would only add zeros)
#pragma unroll
      for (int p = 0; p < kThinPieces; ++p) {
        const int kk = (p * 16 + l16) * 4;
        if (kk < k) {
          f32x4 o;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float gk = d[u][0] * wr[p][t][0] + d[u][1] * wr[p][t][1] + d[u][2] * wr[p][t][2];
            o[t] = (!apply_mask || xv[u][p][t] > 0.f) ? gk : 0.f;
            dwp[p][t][0] += xv[u][p][t] * d[u][0];
            dwp[p][t][1] += xv[u][p][t] * d[u][1];
            dwp[p][t][2] += xv[u][p][t] * d[u][2];
          }
          if (gq && kk < nq * 4) {
            *reinterpret_cast<f32x4 *>(gqr[u] + (size_t)(kk >> 2) * n_vert * 4) = o;
          } else if (kk + 3 < n_store) {
            *reinterpret_cast<f32x4 *>(gr[u] + kk) = o;
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (kk + t < n_store) gr[u][kk + t] = o[t];
          }
        }
      }
    }
  }
  // Reduce dW partials over the 16 row groups of the block, one l16 column at a time.
  float *slab = dw_slab + (size_t)blockIdx.x * k * 3;
  for (int col = 0; col < 16; ++col) {
    if (l16 == col) {
#pragma unroll
      for (int p = 0; p < kThinPieces; ++p)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 3; ++j) red[grp][(p * 4 + t) * 3 + j] = dwp[p][t][j];
    }
    __syncthreads();
    if (threadIdx.x < kThinPieces * 12) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += red[r][threadIdx.x];
      const int p = threadIdx.x / 12, t = (threadIdx.x % 12) / 3, j = threadIdx.x % 3;
      const int kk = (p * 16 + col) * 4 + t;
      if (kk < k) slab[kk * 3 + j] = s;
    }
    __syncthreads();
  }
  // bias gradient partial
  if (l16 == 0) {
    red[grp][0] = db0;
    red[grp][1] = db1;
    red[grp][2] = db2;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[r][threadIdx.x];
    db_slab[(size_t)blockIdx.x * 3 + threadIdx.x] = s;
  }
}

__global__ void pad3to4_kernel(const float *__restrict__ in, long long m, float *__restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  *reinterpret_cast<f32x4 *>(out + i * 4) = f32x4{in[i * 3], in[i * 3 + 1], in[i * 3 + 2], 0.f};
}
int launch_pad3to4(const float *in, long long m, float *out, hipStream_t s) {
  A3VT_LAUNCH(pad3to4_kernel, dim3(cdiv(m, 256)), dim3(256), 0, s, in, m, out);
  A3VT_CHECK_LAUNCH();
  return 0;
}

int launch_thin_bwd(const float *x, int ldx, int k, const float *w, const int32_t *rowptrT, const int32_t *colT,
                    const float *valT, const int32_t *heavyT, int n_vert, int batch, const float *grad_update,
                    float *dz3, int apply_mask, float *g_prev, int ldg, int n_store, float *dw_slab, float *db_slab,
                    float *gq, int nq, const float *xq, int xq_quads, hipStream_t s, const SplitRef *sp) {
  if (k > kThinPieces * 64 || ldx % 4 != 0 || ldg % 4 != 0) {
    set_error("thin_bwd: k=%d ldx=%d ldg=%d unsupported", k, ldx, ldg);
    return -1;
  }
  const long long m = (long long)batch * n_vert;
  // dz3 = A^T dU.  csr3_kernel gathers float4 rows, so dU [M][3] is first padded to [M][4].
  // The dz3 scratch is [2][M][4]: first half = padded dU, second half = A^T dU.
  float *du4 = dz3;
  float *res = dz3 + m * 4;
  if (int rc = launch_pad3to4(grad_update, m, du4, s)) return rc;
  if (int rc = launch_csr3(du4, nullptr, rowptrT, colT, valT, heavyT, n_vert, batch, res, 4, s, sp, true)) return rc;
  A3VT_LAUNCH(thin_bwd_kernel, dim3(kThinBlocks), dim3(256), 0, s, x, ldx, k, w, res, grad_update, m, apply_mask,
                     g_prev, ldg, n_store, dw_slab, db_slab, gq, nq, n_vert, xq, xq_quads);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Vertex update (model.py:250,270,283), finite check (replaces the blocking NaN trap at model.py:326), fill.
// ------------------------------------------------------------------------------------------------
__global__ void vertex_update_kernel(const float *__restrict__ vin, const float *__restrict__ upd, long long total,
                                     int n_vert, int n_vision, float *__restrict__ vout) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int v = (int)((i / 3) % n_vert);
  vout[i] = v < n_vision ? vin[i] + upd[i] : vin[i];
}

int launch_vertex_update(const float *vin, const float *upd, int batch, int n_vert, int n_vision, float *vout,
                         hipStream_t s) {
  const long long total = (long long)batch * n_vert * 3;
  A3VT_LAUNCH(vertex_update_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, vin, upd, total, n_vert, n_vision,
                     vout);
  A3VT_CHECK_LAUNCH();
  return 0;
}

__global__ void check_finite_kernel(const float *__restrict__ d, size_t n, int32_t *flag) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = d[i];
    bad |= !(v - v == 0.f);  // NaN or +-Inf
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

int launch_check_finite(const float *d, size_t n, int32_t *flag, hipStream_t s) {
  const int grid = (int)(cdiv((long long)n, 256) < 2048 ? cdiv((long long)n, 256) : 2048);
  A3VT_LAUNCH(check_finite_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, s, d, n, flag);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// Stand-alone layer backward: gradient through the activation, split into the gather input and the merged dZ rows.
__global__ void relu_split_kernel(const float *__restrict__ gy, int ldgy, const float *__restrict__ y, int ldy,
                                  int relu, int n_out, int cpad, int npad, long long m, float *__restrict__ ga,
                                  float *__restrict__ dz) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * npad) return;
  const long long r = i / npad;
  const int c = (int)(i - r * npad);
  float v = 0.f;
  if (c < n_out) {
    v = gy[r * ldgy + c];
    if (relu && !(y[r * ldy + c] > 0.f)) v = 0.f;
  }
  if (c < cpad) ga[r * cpad + c] = v;
  else dz[r * npad + c] = v;
}
int launch_relu_split(const float *gy, int ldgy, const float *y, int ldy, int relu, int n_out, int cpad, int npad,
                      long long m, float *ga, float *dz, hipStream_t s) {
  A3VT_LAUNCH(relu_split_kernel, dim3((unsigned)cdiv(m * npad, 256)), dim3(256), 0, s, gy, ldgy, y, ldy, relu, n_out,
              cpad, npad, m, ga, dz);
  A3VT_CHECK_LAUNCH();
  return 0;
}

__global__ void fill_zero_kernel(float *__restrict__ d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = 0.f;
}
int launch_fill_zero(float *d, size_t n, hipStream_t s) {
  const int grid = (int)(cdiv((long long)n, 256) < 2048 ? cdiv((long long)n, 256) : 2048);
  A3VT_LAUNCH(fill_zero_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, s, d, n);
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace a3vt
