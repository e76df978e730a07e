// gcn_csrqs.hip — channel-sliced neighbour aggregation for STRUCTURED adjacencies D^-1 (P + J) (gfx950, round 6).
//
// The fused vision + touch adjacency of utility/utils.py:75-130 links EVERY seam vertex of the chart atlas (vertices that share
// a position with another one, :80-84,119-123) to EVERY touch-chart centre (:95-98,124-128): the matrix the layers multiply
// with (reconstruction/vision/model.py:356,360) is
//     A^ = D^-1 (P + J),   P = a sparse symmetric 0/1 pattern (mesh edges, self loops, seam cliques, chart edges; <= 10 per row),
//                          J = the COMPLETE bipartite block S x C  (S = seam vertices, C = chart centres; 75 % of the non-zeros
//                              of the 20-chart graph), D = row sums of P + J.
// So   (A^ Z)_i = (1/d_i) ( sum_{j in P(i)} Z_j + [i in S] sigma_C + [i in C] sigma_S ),  sigma_X = sum_{x in X} Z_x,
// and  (A^T G)_j = sum_{i in P(j)} G_i/d_i + [j in S] sum_{c in C} G_c/d_c + [j in C] sum_{s in S} G_s/d_s   (P, J symmetric):
// two per-mesh sums replace the 1153-entry hub rows and the 26-30-entry seam rows, and what is left has the degree of a mesh.
// a3vt_adj_split (include/a3vt.h) carries the decomposition; a3vt_adj_split_validate proves it against the full CSR.
//
// Kernel = the design of csrq_kernel (gcn_csrq.hip: one persistent workgroup per (mesh, part of the channel quads), the mesh's
// slice of a quad resident in LDS, double buffered, one barrier per quad, quad-major inputs and outputs) with
//   * unit weights: the slice is scaled on its way INTO LDS in the backward (G_v / d_v) and the sum is scaled on its way OUT in
//     the forward, so a thread keeps only COLUMNS for its vertices: 12 slots as six registers of two 16-bit LDS byte offsets
//     (csrq: 8 slots = 16 registers), empty slots point at a row of zeros; slot pairs beyond the longest row of a wave's 64
//     vertices are skipped by a scalar branch;
//   * sigma_S, sigma_C of a quad formed while its slice is parked: per-thread partial sums, a DPP sum over each 16-lane row,
//     32 row sums per class in LDS, and — behind the barrier the slice needs anyway — one lane-parallel read and a DPP tree per
//     wave.  Fixed order everywhere: outputs are bit-repeatable and do not depend on the batch size or on `parts`.
// Not bit-identical to the row walk over the full CSR (other association of the same sum): gated by the fp64 oracle.
#include "common.h"
#include "kernels.h"

namespace a3vt {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kQsThreads = 512;
constexpr int kQsPairs = 6;              // 12 edge slots per vertex
constexpr int kQsWaves = kQsThreads / 64;

int csrqs_max_degree() { return 2 * kQsPairs; }
size_t csrqs_image_ints(int n_vert) { return (size_t)n_vert * (kQsPairs + 2) + 64; }
bool csrqs_fits(int n_vert, int cut_len) {
  // two slices of one mesh in LDS, 16-bit byte offsets of the rows (the row of zeros included), <= 6 vertices per thread
  return cut_len > 0 && pad4(cut_len) <= 128 && n_vert <= 6 * kQsThreads && ((size_t)n_vert + 1) * 16 <= 65535 &&
         ((size_t)n_vert + 1) * 32 <= 158 * 1024;
}

// Index image, built once per stack call from the split (n_vert x 8 words):
//   img[jp * n_vert + v]        two LDS byte offsets (16 * column) of edges 2 jp, 2 jp + 1 of row v of P, ascending columns;
//                               past the row's end: 16 * n_vert = the row of zeros behind the slice
//   img[6 * n_vert + v]         bits 0..7 edge count, bits 8..9 class (0, 1 = S, 2 = C)
//   img[7 * n_vert + v]         float bits of 1 / d_v
__global__ void csrqs_image_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                   const float *__restrict__ scale, const uint8_t *__restrict__ cls, int n_vert,
                                   uint32_t *__restrict__ img) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_vert) return;
  const int e0 = rowptr[v], n = rowptr[v + 1] - e0;
  const uint32_t zero_row = (uint32_t)n_vert * 16u;
#pragma unroll
  for (int jp = 0; jp < kQsPairs; ++jp) {
    const uint32_t a = 2 * jp < n ? (uint32_t)colidx[e0 + 2 * jp] * 16u : zero_row;
    const uint32_t b = 2 * jp + 1 < n ? (uint32_t)colidx[e0 + 2 * jp + 1] * 16u : zero_row;
    img[(size_t)jp * n_vert + v] = a | (b << 16);
  }
  img[(size_t)kQsPairs * n_vert + v] = (uint32_t)(n < 255 ? n : 255) | ((uint32_t)(cls[v] & 3u) << 8);
  img[(size_t)(kQsPairs + 1) * n_vert + v] = __builtin_bit_cast(uint32_t, scale[v]);
}
int launch_csrqs_image(const int32_t *rowptr, const int32_t *col, const float *scale, const uint8_t *cls, int n_vert,
                       int32_t *img, hipStream_t s) {
  A3VT_LAUNCH(csrqs_image_kernel, dim3(cdiv(n_vert, 256)), dim3(256), 0, s, rowptr, col, scale, cls, n_vert,
              reinterpret_cast<uint32_t *>(img));
  A3VT_CHECK_LAUNCH();
  return 0;
}

template <unsigned CTRL>
__device__ __forceinline__ float qs_dpp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row; every lane of the row gets it (a fixed tree: quad, quad pairs, row)
__device__ __forceinline__ float qs_row_sum(float x) {
  x += qs_dpp<0xB1>(x);    // quad_perm [1,0,3,2]
  x += qs_dpp<0x4E>(x);    // quad_perm [2,3,0,1]
  x += qs_dpp<0x124>(x);   // row_ror:4
  x += qs_dpp<0x128>(x);   // row_ror:8
  return x;
}
__device__ __forceinline__ f32x4 qs_row_sum4(f32x4 x) {
  return f32x4{qs_row_sum(x[0]), qs_row_sum(x[1]), qs_row_sum(x[2]), qs_row_sum(x[3])};
}
// lanes 0..31 of a wave hold one partial each (lanes 32..63: copies): the sum of the 32, wave-uniform
__device__ __forceinline__ float qs_sum32(float x) {
  x = qs_row_sum(x);
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 16));
  return a + b;
}
__device__ __forceinline__ int qs_wave_max(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

// MODE 0: Y = relu(scale * (P Z + bipartite) + bias) on the aggregated channels (model.py:356-358,363), sign bytes out.
// MODE 1: dZa = P (G .* sign / d) + bipartite, bias-gradient partial per (mesh, quad) (autograd of the same lines).
template <int MODE, int VPT>
__global__ __launch_bounds__(kQsThreads) void csrqs_kernel(const float *__restrict__ srcq, int nq, int parts,
                                                           const float *__restrict__ bias, int c,
                                                           const uint32_t *__restrict__ img, int n_vert, int batch,
                                                           float *__restrict__ dst, int nq_dst,
                                                           uint8_t *__restrict__ signq, int relu,
                                                           float *__restrict__ db_slab) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  // row sums of the per-thread partials: [parity][class S, C, bias gradient][wave * 4 + DPP row]
  __shared__ f32x4 red[2][3][kQsWaves * 4];
  f32x4 *tile0 = reinterpret_cast<f32x4 *>(lds_raw);
  // the parts of a mesh share an XCD (blockIdx % 8), as csrq_kernel
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int jm = loc / parts, part = loc - jm * parts;
  const int b = xcd + 8 * jm;
  if (b >= batch) return;   // uniform per workgroup
  const int q_lo = part * nq / parts, q_hi = (part + 1) * nq / parts;
  if (q_lo >= q_hi) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  // this thread's vertices, held for all quads.  Threads past the mesh's end work on a copy of the last vertex, belong to no
  // class and skip the stores (branch-free loads, as csrq_kernel).
  int vv[VPT], npairs[VPT];
  bool on[VPT];
  uint32_t pk[VPT][kQsPairs];
  float wv[VPT], fs[VPT], fc[VPT];
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int v = threadIdx.x + k * kQsThreads;
    on[k] = v < n_vert;
    vv[k] = on[k] ? v : n_vert - 1;
    const uint32_t info = img[(size_t)kQsPairs * n_vert + vv[k]];
    const int cl = on[k] ? (int)((info >> 8) & 3u) : 0;
    fs[k] = cl == 1 ? 1.f : 0.f;
    fc[k] = cl == 2 ? 1.f : 0.f;
    wv[k] = __builtin_bit_cast(float, img[(size_t)(kQsPairs + 1) * n_vert + vv[k]]);
    npairs[k] = __builtin_amdgcn_readfirstlane(qs_wave_max((int)((info & 255u) + 1) >> 1));
#pragma unroll
    for (int jp = 0; jp < kQsPairs; ++jp) pk[k][jp] = img[(size_t)jp * n_vert + vv[k]];
  }

  auto fetch = [&](int q, int k, f32x4 &v4, unsigned &bits) {
    const size_t plane = ((size_t)b * nq + q) * n_vert;
    v4 = reinterpret_cast<const f32x4 *>(srcq)[plane + vv[k]];
    bits = MODE == 1 ? signq[plane + vv[k]] : 0u;
  };
  // a slice element on its way to LDS.  MODE 1: the gradient passes the ReLU (all four channels of the quad: the forward
  // kept a sign bit for the pass-through ones too), feeds the bias gradient and is divided by its row's degree — the
  // pass-through channels of the last quad (>= c, model.py:358) stay as they are: they are copied, not aggregated.
  auto park = [&](f32x4 *tile, int q, int k, f32x4 v4, unsigned bits, f32x4 &ps, f32x4 &pc, f32x4 &pb) {
    if (MODE == 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        v4[t] = ((bits >> t) & 1u) ? v4[t] : 0.f;
        pb[t] += on[k] ? v4[t] : 0.f;
        if (q * 4 + t < c) v4[t] *= wv[k];
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ps[t] = __builtin_fmaf(fs[k], v4[t], ps[t]);
      pc[t] = __builtin_fmaf(fc[k], v4[t], pc[t]);
    }
    tile[vv[k]] = v4;   // (the copies of the last vertex all write the same value)
  };
  auto publish = [&](int par, f32x4 ps, f32x4 pc, f32x4 pb) {
    ps = qs_row_sum4(ps);
    pc = qs_row_sum4(pc);
    if (MODE == 1) pb = qs_row_sum4(pb);
    if ((lane & 15) == 0) {
      const int slot = wave * 4 + (lane >> 4);
      red[par][0][slot] = ps;
      red[par][1][slot] = pc;
      if (MODE == 1) red[par][2][slot] = pb;
    }
  };

  if (threadIdx.x < 2) tile0[(size_t)threadIdx.x * (n_vert + 1) + n_vert] = f32x4{0.f, 0.f, 0.f, 0.f};   // the rows of zeros
  {
    f32x4 v4[VPT];
    unsigned bits[VPT];
    f32x4 ps = {0.f, 0.f, 0.f, 0.f}, pc = ps, pb = ps;
#pragma unroll
    for (int k = 0; k < VPT; ++k) fetch(q_lo, k, v4[k], bits[k]);
#pragma unroll
    for (int k = 0; k < VPT; ++k) park(tile0, q_lo, k, v4[k], bits[k], ps, pc, pb);
    publish(0, ps, pc, pb);
  }
  __syncthreads();

  for (int q = q_lo; q < q_hi; ++q) {
    const int par = (q - q_lo) & 1;
    const char *tile = reinterpret_cast<const char *>(tile0 + (size_t)par * (n_vert + 1));
    f32x4 *tnext = tile0 + (size_t)(par ^ 1) * (n_vert + 1);
    const int ch = q * 4;
    const size_t plane = ((size_t)b * nq + q) * n_vert;
    const int qn = q + 1 < q_hi ? q + 1 : q;   // last quad: re-fetch itself (parked nowhere)
    f32x4 nv4[VPT];
    unsigned nbits[VPT];
#pragma unroll
    for (int k = 0; k < VPT; ++k) fetch(qn, k, nv4[k], nbits[k]);

    // this quad's class sums: 32 row sums each, one lane-parallel read per class, a DPP tree, wave-uniform results
    f32x4 sig_s, sig_c;
    {
      const f32x4 a = red[par][0][lane & 31], d = red[par][1][lane & 31];
#pragma unroll
      for (int t = 0; t < 4; ++t) sig_s[t] = qs_sum32(a[t]), sig_c[t] = qs_sum32(d[t]);
    }
    if (MODE == 1 && wave == 0) {   // bias-gradient partial of this (mesh, quad)
      const f32x4 a = red[par][2][lane & 31];
      f32x4 o;
#pragma unroll
      for (int t = 0; t < 4; ++t) o[t] = qs_sum32(a[t]);
      if (lane == 0) *reinterpret_cast<f32x4 *>(db_slab + (size_t)b * (nq * 4) + ch) = o;
    }
    f32x4 bs = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (ch + t < c) bs[t] = bias[ch + t];
    }
    const bool tailq = ch + 4 > c;   // the quad holds pass-through channels (uniform)
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      // one vertex at a time (the scheduler would otherwise hoist the LDS gathers of all the thread's vertices)
      __builtin_amdgcn_sched_barrier(0);
      const int v = vv[k];
      f32x4 own = {0.f, 0.f, 0.f, 0.f};
      if (tailq) own = *reinterpret_cast<const f32x4 *>(tile + (size_t)v * 16);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jp = 0; jp < kQsPairs; ++jp) {
        if (jp < npairs[k]) {   // scalar branch: no row of this wave's 64 vertices is longer
          const f32x4 r0 = *reinterpret_cast<const f32x4 *>(tile + (pk[k][jp] & 0xffffu));
          const f32x4 r1 = *reinterpret_cast<const f32x4 *>(tile + (pk[k][jp] >> 16));
          acc[0] += r0[0]; acc[1] += r0[1]; acc[2] += r0[2]; acc[3] += r0[3];
          acc[0] += r1[0]; acc[1] += r1[1]; acc[2] += r1[2]; acc[3] += r1[3];
        }
      }
      // the bipartite block: a seam vertex receives the centres' sum, a centre the seam vertices' sum
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] += __builtin_fmaf(fs[k], sig_c[t], fc[k] * sig_s[t]);
      if (!on[k]) continue;
      f32x4 o;
      if (MODE == 0) {
        unsigned bits = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // aggregated channels: scaled neighbour sum + bias; pass-through channels of the last quad (model.py:358: no
          // bias): the raw value itself
          const float pre = ch + t < c ? __builtin_fmaf(wv[k], acc[t], bs[t]) : own[t];
          o[t] = (pre > 0.f || !relu) ? pre : 0.f;
          bits |= (pre > 0.f ? 1u : 0u) << t;
        }
        if (signq) signq[plane + v] = (uint8_t)bits;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = ch + t < c ? acc[t] : own[t];
      }
      *reinterpret_cast<f32x4 *>(dst + ((((size_t)b * nq_dst + q) * n_vert) + v) * 4) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (q + 1 < q_hi) {
      f32x4 ps = {0.f, 0.f, 0.f, 0.f}, pc = ps, pb = ps;
#pragma unroll
      for (int k = 0; k < VPT; ++k) park(tnext, q + 1, k, nv4[k], nbits[k], ps, pc, pb);
      publish(par ^ 1, ps, pc, pb);
    }
    __syncthreads();   // next slice and its row sums visible; everyone is done with this slice and with red[par]
  }
}

template <int MODE, int VPT>
static int launch_csrqs_vpt(const float *srcq, int nq, const float *bias, int c, const int32_t *img, int n_vert, int batch,
                            float *dst, int nq_dst, uint8_t *signq, int relu, float *db_slab, hipStream_t s) {
  const size_t shmem = ((size_t)n_vert + 1) * 32;
  static OncePerDevice once;
  once.run([] {
    (void)hipFuncSetAttribute((const void *)csrqs_kernel<MODE, VPT>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
  });
  // one workgroup per CU when the batch allows it: parts of a mesh = 256 / batch (at least 1, at most one per quad)
  int parts = 256 / (((batch + 7) / 8) * 8);
  parts = parts < 1 ? 1 : parts > nq ? nq : parts;
  const int grid = 8 * ((batch + 7) / 8) * parts;   // (XCD group, mesh of the group, part)
  A3VT_LAUNCH((csrqs_kernel<MODE, VPT>), dim3(grid), dim3(kQsThreads), shmem, s, srcq, nq, parts, bias, c,
              reinterpret_cast<const uint32_t *>(img), n_vert, batch, dst, nq_dst, signq, relu, db_slab);
  A3VT_CHECK_LAUNCH();
  return 0;
}

template <int MODE>
static int launch_csrqs(const float *srcq, const float *bias, int c, const int32_t *img, int n_vert, int batch, float *dst,
                        int nq_dst, uint8_t *signq, int relu, float *db_slab, hipStream_t s) {
  const int nq = pad4(c) / 4;
  if (!csrqs_fits(n_vert, c) || nq_dst < nq || (MODE == 1 && (!signq || !db_slab))) {
    set_error("csrqs: n_vert=%d c=%d output planes=%d unsupported", n_vert, c, nq_dst);
    return -1;
  }
  if (n_vert <= 4 * kQsThreads)
    return launch_csrqs_vpt<MODE, 4>(srcq, nq, bias, c, img, n_vert, batch, dst, nq_dst, signq, relu, db_slab, s);
  return launch_csrqs_vpt<MODE, 6>(srcq, nq, bias, c, img, n_vert, batch, dst, nq_dst, signq, relu, db_slab, s);
}

int launch_csrqs_fwd(const float *zq, const float *bias, int c, const int32_t *img, int n_vert, int batch, float *yq,
                     int yq_quads, uint8_t *signq, int relu, hipStream_t s) {
  return launch_csrqs<0>(zq, bias, c, img, n_vert, batch, yq, yq_quads, signq, relu, nullptr, s);
}
int launch_csrqs_bwd(const float *gq, int c, const int32_t *img, int n_vert, int batch, float *dzaq, const uint8_t *signq,
                     float *db_slab, hipStream_t s) {
  return launch_csrqs<1>(gq, nullptr, c, img, n_vert, batch, dzaq, pad4(c) / 4, const_cast<uint8_t *>(signq), 0, db_slab, s);
}

}  // namespace a3vt
